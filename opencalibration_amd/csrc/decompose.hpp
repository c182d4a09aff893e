// cv::decomposeHomographyMat(H, K = I) and the pose list of homography_model::decompose
// (src/model_inliers/homography_model.cpp:138-185), shared by the host classes (g++) and the device (hipcc), both built
// with -ffp-contract=off so that the two agree to the bit.
//
// The decomposition is the analytical one of Malis & Vargas (INRIA RR-6303) as implemented by OpenCV's
// HomographyDecompInria: a third-party algorithm restated from its publication; results are pinned by the reference's
// test/test_ransac_unit.cpp tolerances (tests/test_oracle_ransac.py) and against the oracle's restatement.
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#define OCHIP_DC __host__ __device__ inline
#else
#define OCHIP_DC inline
#endif

namespace ochip_dc
{

struct M3
{
    double a[3][3];
};

OCHIP_DC M3 mul(const M3 &x, const M3 &y)
{
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            r.a[i][j] = x.a[i][0] * y.a[0][j] + x.a[i][1] * y.a[1][j] + x.a[i][2] * y.a[2][j];
    return r;
}
OCHIP_DC M3 transposed(const M3 &x)
{
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            r.a[i][j] = x.a[j][i];
    return r;
}
OCHIP_DC double det3(const M3 &m)
{
    return m.a[0][0] * (m.a[1][1] * m.a[2][2] - m.a[1][2] * m.a[2][1]) -
           m.a[0][1] * (m.a[1][0] * m.a[2][2] - m.a[1][2] * m.a[2][0]) +
           m.a[0][2] * (m.a[1][0] * m.a[2][1] - m.a[1][1] * m.a[2][0]);
}

// middle eigenvalue of a symmetric positive semi-definite 3x3 via cyclic Jacobi rotations
OCHIP_DC double middle_eigenvalue(M3 s)
{
    for (int sweep = 0; sweep < 64; sweep++)
    {
        if (s.a[0][1] == 0 && s.a[0][2] == 0 && s.a[1][2] == 0)
            break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++)
            {
                const double apq = s.a[p][q];
                if (apq == 0)
                    continue;
                const double theta = (s.a[q][q] - s.a[p][p]) / (2 * apq);
                const double t = copysign(1.0, theta) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), sn = t * c;
                for (int k = 0; k < 3; k++)
                {
                    const double x = s.a[k][p], y = s.a[k][q];
                    s.a[k][p] = c * x - sn * y;
                    s.a[k][q] = sn * x + c * y;
                }
                for (int k = 0; k < 3; k++)
                {
                    const double x = s.a[p][k], y = s.a[q][k];
                    s.a[p][k] = c * x - sn * y;
                    s.a[q][k] = sn * x + c * y;
                }
            }
    }
    double a = s.a[0][0], b = s.a[1][1], c = s.a[2][2], t; // the middle of the three (what sorting them leaves in the middle)
    if (a > b)
        t = a, a = b, b = t;
    if (b > c)
        t = b, b = c, c = t;
    if (a > b)
        t = a, a = b, b = t;
    return b;
}

OCHIP_DC double opp_minor(const M3 &m, int row, int col)
{
    const int x1 = col == 0 ? 1 : 0, x2 = col == 2 ? 1 : 2, y1 = row == 0 ? 1 : 0, y2 = row == 2 ? 1 : 2;
    return m.a[y1][x2] * m.a[y2][x1] - m.a[y1][x1] * m.a[y2][x2];
}
OCHIP_DC int sgn(double x)
{
    return x >= 0 ? 1 : -1;
}

struct motion
{
    M3 R;
    double t[3], n[3];
};

OCHIP_DC M3 rotation_from_tstar_n(const M3 &Hn, const double ts[3], const double n[3], double v)
{
    M3 m;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            m.a[i][j] = (i == j ? 1.0 : 0.0) - (2 / v) * ts[i] * n[j];
    M3 R = mul(Hn, m);
    if (det3(R) < 0)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                R.a[i][j] *= -1;
    return R;
}

OCHIP_DC int decompose_homography(const double H[9], motion out[4])
{
    M3 h;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            h.a[i][j] = H[3 * i + j];
    const double ev = middle_eigenvalue(mul(transposed(h), h));
    const double sv = sqrt(ev > 0 ? ev : 0.0);
    M3 Hn;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            Hn.a[i][j] = h.a[i][j] * (1.0 / sv);

    M3 S = mul(transposed(Hn), Hn);
    for (int i = 0; i < 3; i++)
        S.a[i][i] -= 1.0;
    double ninf = 0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            ninf = fmax(ninf, fabs(S.a[i][j]));
    if (ninf < 0.001) // pure rotation
    {
        out[0].R = Hn;
        for (int i = 0; i < 3; i++)
            out[0].t[i] = out[0].n[i] = 0;
        return 1;
    }
    const double M00 = opp_minor(S, 0, 0), M11 = opp_minor(S, 1, 1), M22 = opp_minor(S, 2, 2);
    const double r00 = sqrt(M00), r11 = sqrt(M11), r22 = sqrt(M22);
    const int e12 = sgn(opp_minor(S, 1, 2)), e02 = sgn(opp_minor(S, 0, 2)), e01 = sgn(opp_minor(S, 0, 1));
    const double n0 = fabs(S.a[0][0]), n1 = fabs(S.a[1][1]), n2 = fabs(S.a[2][2]);
    int idx = 0;
    if (n0 < n1)
    {
        idx = 1;
        if (n1 < n2)
            idx = 2;
    }
    else if (n0 < n2)
        idx = 2;
    double pa[3], pb[3];
    if (idx == 0)
    {
        pa[0] = pb[0] = S.a[0][0];
        pa[1] = S.a[0][1] + r22, pb[1] = S.a[0][1] - r22;
        pa[2] = S.a[0][2] + e12 * r11, pb[2] = S.a[0][2] - e12 * r11;
    }
    else if (idx == 1)
    {
        pa[0] = S.a[0][1] + r22, pb[0] = S.a[0][1] - r22;
        pa[1] = pb[1] = S.a[1][1];
        pa[2] = S.a[1][2] - e02 * r00, pb[2] = S.a[1][2] + e02 * r00;
    }
    else
    {
        pa[0] = S.a[0][2] + e01 * r11, pb[0] = S.a[0][2] - e01 * r11;
        pa[1] = S.a[1][2] + r00, pb[1] = S.a[1][2] - r00;
        pa[2] = pb[2] = S.a[2][2];
    }
    const double tr = S.a[0][0] + S.a[1][1] + S.a[2][2];
    // OpenCV uses sqrtf here; (float)sqrt((double)x) is the correctly rounded float square root (what sqrtf returns on the
    // host) on either side
    const double v = 2.0 * (double)(float)sqrt((double)(float)(1 + tr - M00 - M11 - M22));
    const double es = sgn(S.a[idx][idx]);
    const double r = sqrt(2 + tr + v), nt = sqrt(2 + tr - v);
    const double la = sqrt(pa[0] * pa[0] + pa[1] * pa[1] + pa[2] * pa[2]);
    const double lb = sqrt(pb[0] * pb[0] + pb[1] * pb[1] + pb[2] * pb[2]);
    double na[3], nb[3], tas[3], tbs[3];
    for (int i = 0; i < 3; i++)
    {
        na[i] = pa[i] / la;
        nb[i] = pb[i] / lb;
    }
    const double half_nt = 0.5 * nt, esr = es * r;
    for (int i = 0; i < 3; i++)
    {
        tas[i] = (nb[i] * esr - na[i] * nt) * half_nt;
        tbs[i] = (na[i] * esr - nb[i] * nt) * half_nt;
    }
    const M3 Ra = rotation_from_tstar_n(Hn, tas, na, v), Rb = rotation_from_tstar_n(Hn, tbs, nb, v);
    for (int s = 0; s < 4; s++)
    {
        const M3 &R = s < 2 ? Ra : Rb;
        const double *ts = s < 2 ? tas : tbs, *nn = s < 2 ? na : nb;
        const double sign = (s % 2 == 0) ? 1.0 : -1.0;
        out[s].R = R;
        for (int i = 0; i < 3; i++)
        {
            out[s].t[i] = (R.a[i][0] * ts[0] + R.a[i][1] * ts[1] + R.a[i][2] * ts[2]) * sign;
            out[s].n[i] = nn[i] * sign;
        }
    }
    return 4;
}

// Eigen::Quaterniond(Matrix3d): Shoemake's method, Eigen/src/Geometry/Quaternion.h
OCHIP_DC void quaternion_from_rotation(const M3 &m, double q[4])
{
    double t = m.a[0][0] + m.a[1][1] + m.a[2][2];
    if (t > 0)
    {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m.a[2][1] - m.a[1][2]) * t;
        q[1] = (m.a[0][2] - m.a[2][0]) * t;
        q[2] = (m.a[1][0] - m.a[0][1]) * t;
        return;
    }
    int i = 0;
    if (m.a[1][1] > m.a[0][0])
        i = 1;
    if (m.a[2][2] > m.a[i][i])
        i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m.a[i][i] - m.a[j][j] - m.a[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m.a[k][j] - m.a[j][k]) * t;
    q[j] = (m.a[j][i] + m.a[i][j]) * t;
    q[k] = (m.a[k][i] + m.a[i][k]) * t;
}

// What homography_model::decompose builds before and after its cheirality vote: per solution the plane normal N, R N, the
// translation and the rotation as a quaternion (x y z w)...
struct vote_plan
{
    int solutions;
    double N[4][3], RN[4][3];
    double q[4][4], t[4][3];
};
OCHIP_DC void plan_votes(const double H[9], vote_plan *plan)
{
    motion motions[4];
    plan->solutions = decompose_homography(H, motions);
    for (int i = 0; i < plan->solutions; i++)
    {
        const M3 &R = motions[i].R;
        const double *N = motions[i].n;
        for (int c = 0; c < 3; c++)
        {
            plan->N[i][c] = N[c];
            plan->RN[i][c] = R.a[c][0] * N[0] + R.a[c][1] * N[1] + R.a[c][2] * N[2];
            plan->t[i][c] = motions[i].t[c];
        }
        quaternion_from_rotation(R, plan->q[i]);
    }
}
// ... and the order std::stable_sort(poses, score >=) leaves the four poses in (homography_model.cpp:178-181; libstdc++
// runs its insertion sort on so few elements, restated here because the comparator is not a strict order and the outcome
// is the algorithm's): order[k] = which solution ends up k-th; scores of the absent solutions are -1.
OCHIP_DC void order_by_votes(const int score_in[4], int order[4])
{
    int s[4];
    for (int i = 0; i < 4; i++)
    {
        s[i] = score_in[i];
        order[i] = i;
    }
    auto comp = [&](int a, int b) { return a >= b; }; // comp(p1, p2) = p1.score >= p2.score
    for (int i = 1; i < 4; i++)                         // std::__insertion_sort
    {
        const int vs = s[i], vo = order[i];
        if (comp(vs, s[0]))
        {
            for (int k = i; k > 0; k--)
            {
                s[k] = s[k - 1];
                order[k] = order[k - 1];
            }
            s[0] = vs;
            order[0] = vo;
        }
        else
        {
            int k = i;
            while (comp(vs, s[k - 1])) // __unguarded_linear_insert
            {
                s[k] = s[k - 1];
                order[k] = order[k - 1];
                k--;
            }
            s[k] = vs;
            order[k] = vo;
        }
    }
}

} // namespace ochip_dc
