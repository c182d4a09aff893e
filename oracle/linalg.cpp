// ORACLE — test infrastructure only (see oracle.hpp).  Restated Eigen primitives. [3P]
#include "oracle.hpp"

#include <algorithm>
#include <utility>

namespace oracle
{

Mat3 identity3()
{
    Mat3 I;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            I.m[i][j] = i == j ? 1.0 : 0.0;
    return I;
}

double frobenius(const Mat3 &A)
{
    double s = 0;
    for (int j = 0; j < 3; j++) // column-major storage order, as Eigen's default redux walks it
        for (int i = 0; i < 3; i++)
            s += A.m[i][j] * A.m[i][j];
    return std::sqrt(s);
}

// Eigen/src/LU/InverseImpl.h compute_inverse<MatrixType,ResultType,3>: cofactors of column 0 give the
// determinant, every entry is cofactor(j,i) * (1/det).
static inline double cofactor(const Mat3 &m, int i, int j)
{
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m.m[i1][j1] * m.m[i2][j2] - m.m[i1][j2] * m.m[i2][j1];
}

Mat3 inverse3(const Mat3 &A)
{
    const double c00 = cofactor(A, 0, 0), c10 = cofactor(A, 1, 0), c20 = cofactor(A, 2, 0);
    const double d = c00 * A.m[0][0] + c10 * A.m[1][0] + c20 * A.m[2][0];
    const double invdet = 1.0 / d;
    Mat3 R;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            R.m[r][c] = cofactor(A, c, r) * invdet;
    return R;
}

// Eigen/src/LU/FullPivLU.h computeInPlace + _solve_impl for an (rows x 9) matrix.
void full_piv_lu_solve9(double *A, size_t rows, const double *rhs, double out[9])
{
    const size_t cols = 9;
    const size_t size = std::min(rows, cols);
    std::vector<size_t> rowT(size);
    size_t colT[9];
    size_t nonzero_pivots = size;
    double maxpivot = 0;

    for (size_t k = 0; k < size; k++)
    {
        // pivot search: column by column, strict '>' so the first maximum in that order wins
        size_t br = k, bc = k;
        double biggest = std::abs(A[k * cols + k]);
        for (size_t j = k; j < cols; j++)
            for (size_t i = k; i < rows; i++)
            {
                const double v = std::abs(A[i * cols + j]);
                if (v > biggest)
                {
                    biggest = v;
                    br = i;
                    bc = j;
                }
            }
        if (biggest == 0.0)
        {
            nonzero_pivots = k;
            for (size_t i = k; i < size; i++)
            {
                rowT[i] = i;
                colT[i] = i;
            }
            break;
        }
        if (biggest > maxpivot)
            maxpivot = biggest;
        rowT[k] = br;
        colT[k] = bc;
        if (k != br)
            for (size_t j = 0; j < cols; j++)
                std::swap(A[k * cols + j], A[br * cols + j]);
        if (k != bc)
            for (size_t i = 0; i < rows; i++)
                std::swap(A[i * cols + k], A[i * cols + bc]);
        if (k < rows - 1)
        {
            const double p = A[k * cols + k];
            for (size_t i = k + 1; i < rows; i++)
                A[i * cols + k] /= p;
        }
        if (k < size - 1)
            for (size_t i = k + 1; i < rows; i++)
            {
                const double l = A[i * cols + k];
                for (size_t j = k + 1; j < cols; j++)
                    A[i * cols + j] -= l * A[k * cols + j];
            }
    }

    // rank(): |diag| > |maxpivot| * eps * diagonalSize
    const double premult = std::abs(maxpivot) * (std::numeric_limits<double>::epsilon() * double(size));
    size_t rank = 0;
    for (size_t i = 0; i < nonzero_pivots; i++)
        rank += (std::abs(A[i * cols + i]) > premult) ? 1 : 0;

    for (int i = 0; i < 9; i++)
        out[i] = 0;
    if (rank == 0)
        return;

    // Step 1: c = P * rhs  (row transpositions in the order they were applied)
    std::vector<double> c(rhs, rhs + rows);
    for (size_t k = 0; k < size; k++)
        std::swap(c[k], c[rowT[k]]);
    // Step 2: unit-lower solve on the top smalldim rows (column-oriented forward substitution)
    for (size_t j = 0; j < size; j++)
    {
        const double cj = c[j];
        for (size_t i = j + 1; i < size; i++)
            c[i] -= cj * A[i * cols + j];
    }
    // (rows > cols: c.bottomRows -= lu.bottomRows * c.topRows — never read afterwards)
    // Step 3: upper solve on the top-left rank x rank block (column-oriented back substitution)
    for (size_t jj = rank; jj-- > 0;)
    {
        c[jj] /= A[jj * cols + jj];
        const double cj = c[jj];
        for (size_t i = 0; i < jj; i++)
            c[i] -= cj * A[i * cols + jj];
    }
    // Step 4: undo the column permutation
    size_t perm[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
    for (size_t k = 0; k < size; k++)
        std::swap(perm[k], perm[colT[k]]);
    for (size_t i = 0; i < rank; i++)
        out[perm[i]] = c[i];
}

} // namespace oracle
