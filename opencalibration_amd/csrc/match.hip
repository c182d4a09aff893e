// libochip.so — brute-force 486-bit Hamming 2-NN over image pairs (gfx950).
//
// Replaces the double loop of src/match/match_features.cpp:71-93 for every directed image pair of a
// link batch in one launch.
//
// Mapping (DESIGN.md "match kernel"): one lane owns QPT query descriptors (16 dwords each, resident in
// VGPRs for the whole kernel); the reference descriptor index k is wave-uniform, so the 64-byte
// reference rows are fetched by the scalar unit (s_load_dwordx16 -> SGPRs) and enter the VALU as
// scalar operands: per (query, reference) the vector pipe executes exactly 16 v_xor_b32 +
// 16 v_bcnt_u32_b32 (accumulating form) + 3 ops of top-2 bookkeeping, with no LDS traffic and no
// cross-lane step.  HBM traffic is one read of each image's descriptors per workgroup (L2/K$ absorb
// the re-reads), so the kernel is integer-VALU bound, not HBM bound.
//
// Top-2 semantics of the reference (strict '<' in a sequential scan over k): best = lexicographic
// minimum of (count, k); second = minimum count over the other k (equals best on a tie).  With
// key = count << 20 | k both are order independent: best' = min(best, key),
// second' = median(best, second, key).
#include "ctx.hpp"

#include <algorithm>
#include <unordered_map>
#include <vector>

namespace
{

constexpr int BLOCK = 256;
constexpr uint32_t KEY_SHIFT = 20;
constexpr uint32_t KEY_MASK = (1u << KEY_SHIFT) - 1;

__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// popcount(x) + acc in ONE instruction (the accumulating form of v_bcnt_u32_b32).  Left to itself the compiler counts
// every word separately and sums the sixteen counts with a tree of v_add3_u32: 7-8 more VALU instructions per compare.
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

template <int QPT>
__global__ __launch_bounds__(BLOCK) void hamming_2nn_kernel(const uint32_t *__restrict__ desc,
                                                            const uint64_t *__restrict__ img_off,
                                                            const uint32_t *__restrict__ img_n,
                                                            const ochip_pair *__restrict__ pairs,
                                                            const uint64_t *__restrict__ out_off,
                                                            ochip_match *__restrict__ out, uint32_t chunks_per_pair)
{
    const uint32_t pair = blockIdx.x / chunks_per_pair;
    const uint32_t chunk = blockIdx.x - pair * chunks_per_pair;
    const ochip_pair pr = pairs[pair];
    const uint32_t n1 = img_n[pr.image_1], n2 = img_n[pr.image_2];
    const uint32_t q0 = chunk * (BLOCK * QPT);
    if (q0 >= n1)
        return;
    const uint32_t *__restrict__ Q = desc + img_off[pr.image_1] * 16;
    const uint32_t *__restrict__ R = desc + img_off[pr.image_2] * 16;

    uint32_t q[QPT][16];
    uint32_t best[QPT], second[QPT];
#pragma unroll
    for (int j = 0; j < QPT; j++)
    {
        uint32_t qi = q0 + j * BLOCK + threadIdx.x;
        qi = qi < n1 ? qi : n1 - 1; // tail lanes recompute the last query and do not store
        const uint4 *src = reinterpret_cast<const uint4 *>(Q + (size_t)qi * 16);
#pragma unroll
        for (int v = 0; v < 4; v++)
        {
            const uint4 t = src[v];
            q[j][4 * v + 0] = t.x;
            q[j][4 * v + 1] = t.y;
            q[j][4 * v + 2] = t.z;
            q[j][4 * v + 3] = t.w;
        }
        best[j] = 0xFFFFFFFFu;
        second[j] = 0xFFFFFFFFu;
    }

    // Software-pipelined over k with two SGPR row buffers: the scalar load of row k+1 is in flight
    // while the VALU works on row k (wave-uniform addresses => s_load_dwordx16).
    auto load_row = [&](uint32_t (&rw)[16], uint32_t k) {
        const uint32_t kk = k < n2 ? k : n2 - 1;
        const uint32_t *__restrict__ r = R + (size_t)kk * 16;
#pragma unroll
        for (int w = 0; w < 16; w++)
            rw[w] = r[w];
    };
    auto score_row = [&](const uint32_t (&rw)[16], uint32_t k) {
#pragma unroll
        for (int j = 0; j < QPT; j++)
        {
            uint32_t cnt = 0;
#pragma unroll
            for (int w = 0; w < 16; w++)
                cnt = bcnt_acc(q[j][w] ^ rw[w], cnt);
            const uint32_t key = (cnt << KEY_SHIFT) | k;
            second[j] = med3_u32(best[j], second[j], key);
            best[j] = best[j] < key ? best[j] : key;
        }
    };
    if (n2 > 0)
    {
        uint32_t ra[16], rb[16];
        load_row(ra, 0);
        uint32_t k = 0;
        for (; k + 1 < n2; k += 2)
        {
            load_row(rb, k + 1);
            score_row(ra, k);
            load_row(ra, k + 2);
            score_row(rb, k + 1);
        }
        if (k < n2)
            score_row(ra, k);
    }

    ochip_match *__restrict__ o = out + out_off[pair];
#pragma unroll
    for (int j = 0; j < QPT; j++)
    {
        const uint32_t qi = q0 + j * BLOCK + threadIdx.x;
        if (qi < n1)
        {
            ochip_match m;
            m.best_k = best[j] & KEY_MASK;
            m.best_count = (uint16_t)(best[j] >> KEY_SHIFT);
            m.second_count = second[j] == 0xFFFFFFFFu ? (uint16_t)OCHIP_NO_SECOND : (uint16_t)(second[j] >> KEY_SHIFT);
            o[qi] = m;
        }
    }
}

// ---- both directions of an image pair from ONE pass over its distance matrix ---------------------------------------
// The link stage asks for (A -> B) and, nearly always, (B -> A): the same n_A x n_B Hamming distances, reduced along
// rows for one direction and along columns for the other.  The kernel above would compute the matrix twice.  Here a
// wavefront owns 64 queries of A as before (row top-2 in registers) and walks B in tiles of 64 references; the 64 x 64
// counts of a tile also go to LDS (two 16-bit counts per word, rows padded to 65 words: conflict-free both ways), the
// wave then reads them transposed - lane j takes reference j's column - and keeps that column's top-2 over its 64
// queries with the same key arithmetic (key = count << 20 | query index, so the lowest query wins ties, exactly what
// the scan of (B -> A) over A's descriptors does).  Every wave writes one partial (best, second) per reference and tile
// - the waves never wait for each other -, and sym_merge_kernel folds the n_A / 64 partials of every reference into
// the (B -> A) records.  Cost per tile and wave: 64 x 35 VALU for the counts as before, plus ~320 instructions for
// the transposed pass: 15 % on top instead of 100 %.  A slot without a distance (query beyond n_A, reference beyond
// n_B) holds 0xFFFF, which the key arithmetic turns into a count of 0xFFF - larger than any real one.
struct sym_job
{
    uint32_t a, b;           // images
    uint64_t off_ab, off_ba; // output offsets of (a -> b) and (b -> a)
    uint64_t part_off;       // first partial of this job: [tile of 64 queries of a][reference of b]
};

__device__ __forceinline__ void top2_merge(uint32_t &b, uint32_t &s, uint32_t ob, uint32_t os)
{
    const uint32_t lo = b < ob ? b : ob, hi = b < ob ? ob : b, ms = s < os ? s : os;
    b = lo;
    s = hi < ms ? hi : ms;
}

__global__ __launch_bounds__(BLOCK) void hamming_2nn_sym_kernel(const uint32_t *__restrict__ desc, const uint64_t *__restrict__ img_off,
                                                                const uint32_t *__restrict__ img_n, const sym_job *__restrict__ jobs,
                                                                ochip_match *__restrict__ out, uint2 *__restrict__ part,
                                                                uint32_t chunks_per_pair)
{
    // per wave: counts of a 32 (references) x 64 (queries) tile, references 2m and 2m + 1 share a word; 4 KB per wave keeps
    // the occupancy the one-direction kernel has (the scalar reference loads need the other waves to hide behind)
    __shared__ uint32_t T[4][16 * 65];
    const uint32_t pair = blockIdx.x / chunks_per_pair;
    const uint32_t chunk = blockIdx.x - pair * chunks_per_pair;
    const sym_job jb = jobs[pair];
    const uint32_t nA = img_n[jb.a], nB = img_n[jb.b];
    const uint32_t q0 = chunk * BLOCK;
    if (q0 >= nA) // uniform for the workgroup
        return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (q0 + wv * 64 >= nA) // this wave has no query (its row tile does not exist); the waves do not depend on each other
        return;
    const uint32_t *__restrict__ Q = desc + img_off[jb.a] * 16;
    const uint32_t *__restrict__ R = desc + img_off[jb.b] * 16;
    const uint32_t qi = q0 + threadIdx.x;
    const bool q_valid = qi < nA;
    uint32_t q[16];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(Q + (size_t)(q_valid ? qi : nA - 1) * 16);
#pragma unroll
        for (int v = 0; v < 4; v++)
        {
            const uint4 t = src[v];
            q[4 * v + 0] = t.x;
            q[4 * v + 1] = t.y;
            q[4 * v + 2] = t.z;
            q[4 * v + 3] = t.w;
        }
    }
    uint32_t best = 0xFFFFFFFFu, second = 0xFFFFFFFFu;
    auto load_row = [&](uint32_t (&rw)[16], uint32_t k) {
        const uint32_t kk = k < nB ? k : nB - 1;
        const uint32_t *__restrict__ r = R + (size_t)kk * 16;
#pragma unroll
        for (int w = 0; w < 16; w++)
            rw[w] = r[w];
    };
    auto count_row = [&](const uint32_t (&rw)[16]) {
        uint32_t cnt = 0;
#pragma unroll
        for (int w = 0; w < 16; w++)
            cnt = bcnt_acc(q[w] ^ rw[w], cnt);
        return cnt;
    };
    uint32_t *Tw = T[wv];
    const uint32_t n_tiles = (nB + 31) / 32;
    for (uint32_t jt = 0; jt < n_tiles; jt++)
    {
        const uint32_t k0 = jt * 32;
        uint32_t ra[16], rb[16];
        load_row(ra, k0);
#pragma unroll 4
        for (uint32_t j = 0; j < 32; j += 2)
        {
            load_row(rb, k0 + j + 1);
            const uint32_t c0 = count_row(ra);
            load_row(ra, k0 + j + 2);
            const uint32_t c1 = count_row(rb);
            const uint32_t ka = k0 + j, kb = ka + 1;
            if (ka < nB) // uniform
            {
                const uint32_t key = (c0 << KEY_SHIFT) | ka;
                second = med3_u32(best, second, key);
                best = best < key ? best : key;
            }
            if (kb < nB)
            {
                const uint32_t key = (c1 << KEY_SHIFT) | kb;
                second = med3_u32(best, second, key);
                best = best < key ? best : key;
            }
            // 0xFFFF: no distance here (query beyond n_A or reference beyond n_B)
            const uint32_t s0 = (q_valid && ka < nB) ? c0 : 0xFFFFu, s1 = (q_valid && kb < nB) ? c1 : 0xFFFFu;
            Tw[(j >> 1) * 65 + lane] = s0 | (s1 << 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // transposed pass: lane (j, half) = reference k0 + j over queries half * 32 .. half * 32 + 31 of this wave, ascending;
        // the two halves of a reference are merged across the half-waves
        uint32_t cb = 0xFFFFFFFFu, cs = 0xFFFFFFFFu;
        {
            const uint32_t jj = lane & 31, half = lane >> 5;
            const uint32_t *col = Tw + (jj >> 1) * 65 + half * 32;
            const uint32_t sh = (jj & 1) * 16;
            const uint32_t qbase = q0 + wv * 64 + half * 32;
#pragma unroll 8
            for (uint32_t i = 0; i < 32; i++)
            {
                const uint32_t c = (col[i] >> sh) & 0xFFFFu;
                const uint32_t key = (c << KEY_SHIFT) | (qbase + i); // 0xFFFF << 20 keeps 0xFFF: "no distance" sorts last
                cs = med3_u32(cb, cs, key);
                cb = cb < key ? cb : key;
            }
            const uint32_t ob = (uint32_t)__shfl_xor((int)cb, 32), os = (uint32_t)__shfl_xor((int)cs, 32);
            top2_merge(cb, cs, ob, os);
            const uint32_t ref = k0 + jj;
            if (half == 0 && ref < nB)
                part[jb.part_off + (size_t)(chunk * 4 + wv) * nB + ref] = make_uint2(cb, cs);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // the tile's LDS is rewritten by the next round
        __builtin_amdgcn_wave_barrier();
    }
    if (q_valid)
    {
        ochip_match m;
        m.best_k = best & KEY_MASK;
        m.best_count = (uint16_t)(best >> KEY_SHIFT);
        m.second_count = second == 0xFFFFFFFFu ? (uint16_t)OCHIP_NO_SECOND : (uint16_t)(second >> KEY_SHIFT);
        out[jb.off_ab + qi] = m;
    }
}

// (b -> a) records of the symmetric jobs: per reference of b the top-2 over the partials of a's query chunks
__global__ __launch_bounds__(BLOCK) void sym_merge_kernel(const sym_job *__restrict__ jobs, const uint32_t *__restrict__ img_n,
                                                          const uint2 *__restrict__ part, ochip_match *__restrict__ out)
{
    const sym_job jb = jobs[blockIdx.x];
    const uint32_t nA = img_n[jb.a], nB = img_n[jb.b];
    const uint32_t r = blockIdx.y * BLOCK + threadIdx.x;
    if (r >= nB)
        return;
    uint32_t b = 0xFFFFFFFFu, s = 0xFFFFFFFFu;
    const uint32_t tiles = (nA + 63) / 64;
    for (uint32_t c = 0; c < tiles; c++)
    {
        const uint2 p = part[jb.part_off + (size_t)c * nB + r];
        top2_merge(b, s, p.x, p.y);
    }
    ochip_match m;
    m.best_k = b & KEY_MASK;
    m.best_count = (uint16_t)(b >> KEY_SHIFT);
    m.second_count = (s >> KEY_SHIFT) == 0xFFFu ? (uint16_t)OCHIP_NO_SECOND : (uint16_t)(s >> KEY_SHIFT); // only one query in a
    out[jb.off_ba + r] = m;
}

// ---- the same 2-NN on the matrix cores ---------------------------------------------------------------------------
// popcount(a ^ b) = |a| + |b| - 2 a.b, and a.b of two 486-bit vectors is a small integer: an MFMA over the bits taken as
// numbers computes 1 024 of them per instruction where the vector pipe needs 32 instructions per 64.  The descriptors are
// expanded once to FP4 (E2M1) values 0 / 1 - 256 bytes each - and v_mfma_scale_f32_32x32x64_f8f6f4 (eight per 32 x 32
// tile of distances) accumulates, in fp32 and exactly (every term is an integer below 2^24):
//     key(query, reference k) = (512 - |ref|) * 8192 + (8191 - k) + 16384 * (query . ref)
// - the first two terms enter as the accumulator's initial value, 16 384 is the block scale of the reference operand.
// key / 8192 = 512 - |ref| + 2 q.r = 512 + |q| - distance, so the LARGEST key of a query is its nearest reference, the
// lowest k among equals, and the second largest key carries the second smallest distance - the reference's scan with its
// strict '<' (match_features.cpp:80-92) again; the vector pipe is left with two instructions per distance-lane-register
// (v_med3_f32, v_max_f32), 64 per 2 048 distances, and the result tile never leaves the registers: an MFMA tile has its
// column on the lane, so with the queries as columns every lane folds its own query's 16 references.
// A wavefront keeps 64 queries as B operands in registers (two tiles); the four waves of a workgroup share the reference
// tiles (64 references, 16 KB) through LDS, double-buffered, rows padded so that the 16-byte operand reads are conflict
// free.  References beyond n2 start from -1e9 and never win.  Needs n2 <= 8 192 (k in 13 bits); larger images take the
// popcount kernels above.
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr uint32_t MFMA_MAX_REFS = 8192;
constexpr int MFMA_QT = 2;      // blocks of 32 queries per wavefront (2: 212 VGPRs, two waves per SIMD, 4.4e12 distances/s; 1: 167 VGPRs, three waves, 4.0e12)
constexpr int MFMA_QUERIES = 4 * 32 * MFMA_QT; // per workgroup of four waves

// bits -> FP4: one thread per byte of a descriptor (8 bits -> 8 nibbles of value 0b0010 = 1.0 or 0), a wave per feature
__global__ __launch_bounds__(256) void expand_fp4_kernel(const uint32_t *__restrict__ desc, uint64_t first, uint64_t n,
                                                         uint32_t *__restrict__ fp4, float *__restrict__ negpop,
                                                         uint32_t *__restrict__ pop)
{
    const uint64_t f = first + (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= first + n)
        return;
    const int j = threadIdx.x & 63;
    const uint32_t x = (desc[f * 16 + (j >> 2)] >> (8 * (j & 3))) & 0xFFu;
    uint32_t t = (x | (x << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    fp4[f * 64 + j] = t << 1;
    uint32_t c = (uint32_t)__popc(x);
    for (int off = 32; off >= 1; off >>= 1)
        c += (uint32_t)__shfl_xor((int)c, off);
    if (j == 0)
    {
        pop[f] = c;
        negpop[f] = (float)((512 - (int)c) * 8192);
    }
}

// top-2 bookkeeping on keys: second' = median(best, second, key), best' = max(best, key).  As compiler builtins (v_med3_f32,
// v_max_i32) - NOT as inline asm: the
// accumulators are VGPRs written by the MFMAs just before, and the wait states between a matrix instruction and a vector
// instruction that reads its result are the compiler's to insert, which it cannot do inside an asm statement.
__device__ __forceinline__ float med3_f32(float a, float b, float c)
{
    return __builtin_amdgcn_fmed3f(a, b, c);
}
__device__ __forceinline__ float max_f32(float a, float b)
{
    // as integers: the keys that matter are positive floats, which order like their bit patterns, and every negative value
    // ("no reference here") stays below them - one v_max_i32, where fmaxf is three instructions with its canonicalising
    return __int_as_float(max(__float_as_int(a), __float_as_int(b)));
}

__global__ __launch_bounds__(256) void hamming_2nn_mfma_kernel(const uint4 *__restrict__ fp4, const float *__restrict__ negpop,
                                                               const uint32_t *__restrict__ pop,
                                                               const uint64_t *__restrict__ img_off,
                                                               const uint32_t *__restrict__ img_n,
                                                               const ochip_pair *__restrict__ pairs,
                                                               const uint64_t *__restrict__ out_off,
                                                               ochip_match *__restrict__ out, uint32_t chunks_per_pair)
{
    // reference rows of a tile: [row][16 x 16 bytes], rows 272 bytes apart: the 16 lanes of a ds_read_b128 group then sit 4
    // banks apart (conflict free) and a lane reads its eight operands at immediate offsets from one address.  A tile is 64
    // references, two MFMA row blocks: one workgroup barrier and one round of staging loads per 32 matrix instructions.
    constexpr int TR = 64;
    __shared__ uint4 tileA[2][TR * 17];
    __shared__ float tileC[2][TR]; // the rows' accumulator start: (512 - |ref|) * 8192 + (8191 - k)
    const uint32_t pair = blockIdx.x / chunks_per_pair;
    const uint32_t chunk = blockIdx.x - pair * chunks_per_pair;
    const ochip_pair pr = pairs[pair];
    const uint32_t n1 = img_n[pr.image_1], n2 = img_n[pr.image_2];
    const uint32_t q0 = chunk * MFMA_QUERIES;
    if (q0 >= n1 || n2 == 0) // uniform for the workgroup (n2 == 0: the caller fills "no match")
        return;
    const uint64_t off1 = img_off[pr.image_1], off2 = img_off[pr.image_2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
    // queries of this wave: columns r of its two tiles; lane (r, h) holds the k range [64 s + 32 h, + 32) of step s
    constexpr int QT = MFMA_QT;
    v8i Bq[QT][8];
#pragma unroll
    for (int t = 0; t < QT; t++)
    {
        uint32_t qi = q0 + wv * (32 * QT) + t * 32 + r;
        qi = qi < n1 ? qi : n1 - 1;
        const uint4 *src = fp4 + (off1 + qi) * 16;
#pragma unroll
        for (int sidx = 0; sidx < 8; sidx++)
        {
            const uint4 v = src[2 * sidx + h];
            Bq[t][sidx] = v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
        }
    }
    float best[QT], second[QT];
#pragma unroll
    for (int t = 0; t < QT; t++)
        best[t] = second[t] = -1.0f;
    const uint32_t n_tiles = (n2 + TR - 1) / TR;
    // staging: a tile's rows are consecutive features, 16 KB in one piece; thread tid moves its 16-byte pieces tid + 256 i.
    // Only the last tile can have rows beyond n2: they re-read the last feature and are disabled through tileC.
    const uint4 *gsrc = fp4 + off2 * 16 + tid;
    const float *gneg = negpop + off2 + (tid & 63);
    const int st0 = (tid >> 4) * 17 + (tid & 15); // LDS slot of the first piece; the others lie 16 rows further each
    auto load_tile = [&](uint32_t jt, uint4 (&p)[4], float &c) {
        if ((jt + 1) * TR <= n2) // (uniform)
        {
#pragma unroll
            for (int i = 0; i < 4; i++)
                p[i] = gsrc[(size_t)jt * (TR * 16) + 256 * i];
            c = gneg[jt * TR] + (float)(8191 - (int)(jt * TR + (uint32_t)(tid & 63)));
        }
        else
        {
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const uint32_t k = jt * TR + ((uint32_t)tid >> 4) + 16 * i;
                p[i] = fp4[(off2 + (k < n2 ? k : n2 - 1)) * 16 + (tid & 15)];
            }
            const uint32_t kc = jt * TR + (uint32_t)(tid & 63);
            c = kc < n2 ? negpop[off2 + kc] + (float)(8191 - (int)kc) : -1e9f;
        }
    };
    auto put_tile = [&](int buf, const uint4 (&p)[4], float c) {
#pragma unroll
        for (int i = 0; i < 4; i++)
            tileA[buf][st0 + 16 * 17 * i] = p[i];
        if (tid < TR)
            tileC[buf][tid] = c;
    };
    {
        uint4 p[4];
        float c;
        load_tile(0, p, c);
        put_tile(0, p, c);
    }
    __syncthreads();
    // Two tiles in flight in registers: the loads of tile j + 2 are issued at the start of tile j and stored to LDS at the
    // end of tile j + 1 - two tile times (~4 000 cycles) for a round trip that takes about one of them when every CU asks at
    // once; with one tile ahead the store at the end of a tile waited for loads issued at its beginning.
    uint4 nx[2][4];
    float nc[2] = {0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int i = 0; i < 4; i++)
            nx[q][i] = make_uint4(0, 0, 0, 0);
    if (n_tiles > 1)
        load_tile(1, nx[1], nc[1]);
    auto tile = [&](uint32_t jt, uint4 (&mine)[4], float &mine_c, uint4 (&ahead)[4], float &ahead_c) {
        // `mine`: tile jt + 1, requested one tile ago; `ahead`: the registers tile jt's own data came through, free again
        const int cur = (int)(jt & 1);
        const bool more = jt + 1 < n_tiles;
        if (jt + 2 < n_tiles)
            load_tile(jt + 2, ahead, ahead_c);
        // Four accumulator tiles at once - the tile's two row blocks x the wave's two query blocks -, so that a matrix
        // instruction's successor on the same accumulator comes three instructions later: with two chains the matrix pipe
        // ran at half its rate (a dependent v_mfma_scale waits for its predecessor's result, ~2 issue slots).
        v16f acc[2][QT];
#pragma unroll
        for (int sub = 0; sub < 2; sub++)
        {
            // accumulator start: register i of lane (r, h) is row (i & 3) + 8 (i >> 2) + 4 h of the row block
#pragma unroll
            for (int gidx = 0; gidx < 4; gidx++)
            {
                const float4 c4 = *reinterpret_cast<const float4 *>(&tileC[cur][32 * sub + 8 * gidx + 4 * h]);
                acc[sub][0][4 * gidx + 0] = c4.x;
                acc[sub][0][4 * gidx + 1] = c4.y;
                acc[sub][0][4 * gidx + 2] = c4.z;
                acc[sub][0][4 * gidx + 3] = c4.w;
            }
#pragma unroll
            for (int t = 1; t < QT; t++)
                acc[sub][t] = acc[sub][0];
        }
        typedef int v4i __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
            // the operands of four k-steps of both row blocks are requested together, ahead of their sixteen MFMAs (read
            // one step at a time the matrix pipe waited out an LDS round trip between instructions)
            v4i a[2][4];
#pragma unroll
            for (int sub = 0; sub < 2; sub++)
            {
                const uint4 *arow = &tileA[cur][(32 * sub + r) * 17 + h + 8 * half];
#pragma unroll
                for (int sidx = 0; sidx < 4; sidx++)
                {
                    const uint4 t = arow[2 * sidx];
                    a[sub][sidx] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
                }
            }
            // (an empty statement that needs all eight in registers at once: left alone the compiler sinks every read to its use)
            asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]));
#pragma unroll
            for (int sidx = 0; sidx < 4; sidx++)
#pragma unroll
                for (int sub = 0; sub < 2; sub++)
                {
                    const v8i A = v8i{a[sub][sidx].x, a[sub][sidx].y, a[sub][sidx].z, a[sub][sidx].w, 0, 0, 0, 0};
                    // FP4 both sides (cbsz = blgp = 4); block scales: 2^14 on the references (E8M0 141), 1 on the queries (127)
#pragma unroll
                    for (int t = 0; t < QT; t++)
                        acc[sub][t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, Bq[t][4 * half + sidx], acc[sub][t], 4, 4, 0, 141, 0, 127);
                }
        }
#pragma unroll
        for (int sub = 0; sub < 2; sub++)
#pragma unroll
            for (int i = 0; i < 16; i++)
            {
#pragma unroll
                for (int t = 0; t < QT; t++)
                {
                    second[t] = med3_f32(best[t], second[t], acc[sub][t][i]);
                    best[t] = max_f32(best[t], acc[sub][t][i]);
                }
            }
        if (more)
            put_tile(cur ^ 1, mine, mine_c);
        __syncthreads();
    };
    for (uint32_t jt = 0; jt < n_tiles; jt += 2)
    {
        tile(jt, nx[1], nc[1], nx[0], nc[0]);
        if (jt + 1 < n_tiles)
            tile(jt + 1, nx[0], nc[0], nx[1], nc[1]);
    }
    // the two half-waves hold the same queries' other 16 rows per tile
    ochip_match *__restrict__ o = out + out_off[pair];
#pragma unroll
    for (int t = 0; t < QT; t++)
    {
        const float ob = __shfl_xor(best[t], 32), os = __shfl_xor(second[t], 32);
        const float hi = fmaxf(best[t], ob), lo = fminf(best[t], ob), ms = fmaxf(second[t], os);
        const float b2 = fmaxf(lo, ms);
        const uint32_t qi = q0 + wv * (32 * QT) + t * 32 + r;
        if (h == 0 && qi < n1)
        {
            const uint32_t pq = pop[off1 + qi];
            const int kb = (int)hi; // exact: an integer below 2^24
            ochip_match m;
            m.best_k = 8191u - ((uint32_t)kb & 8191u);
            m.best_count = (uint16_t)(pq + 512u - ((uint32_t)kb >> 13));
            m.second_count = b2 < 0.0f ? (uint16_t)OCHIP_NO_SECOND : (uint16_t)(pq + 512u - ((uint32_t)(int)b2 >> 13));
            o[qi] = m;
        }
    }
}

} // namespace

extern "C"
{

int ochip_match_launch(ochip_ctx *ctx, const ochip_pair *pairs, uint32_t n_pairs, const uint64_t *out_offset,
                       uint64_t out_total)
{
    if (!ctx)
        return OCHIP_EINVAL;
    if (n_pairs && (!pairs || !out_offset))
        return ochip_fail(ctx, OCHIP_EINVAL, "pairs/out_offset is NULL");
    if (!ctx->desc_dev)
        return ochip_fail(ctx, OCHIP_ESTATE, "ochip_descriptors_reserve has not been called");
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    ctx->match_out_total = out_total;
    if (n_pairs == 0)
        return OCHIP_OK;

    uint32_t max_n1 = 0;
    for (uint32_t p = 0; p < n_pairs; p++)
    {
        const ochip_pair &pr = pairs[p];
        if (pr.image_1 >= ctx->n_images || pr.image_2 >= ctx->n_images || !ctx->img_set[pr.image_1] ||
            !ctx->img_set[pr.image_2])
            return ochip_fail(ctx, OCHIP_ESTATE, "pair %u references an image that was not uploaded", p);
        if (ctx->img_n[pr.image_2] > KEY_MASK)
            return ochip_fail(ctx, OCHIP_EINVAL, "image %u has %u descriptors; the kernel supports <= %u", pr.image_2,
                              ctx->img_n[pr.image_2], KEY_MASK);
        if (out_offset[p] + ctx->img_n[pr.image_1] > out_total)
            return ochip_fail(ctx, OCHIP_EINVAL, "out_offset[%u] + n1 exceeds out_total", p);
        max_n1 = ctx->img_n[pr.image_1] > max_n1 ? ctx->img_n[pr.image_1] : max_n1;
    }

    // ---- pairs whose reference image fits the matrix-core kernel's 13-bit index go there, each direction on its own
    static const bool use_mfma = !ochip_test_hook("popcount_match");
    std::vector<ochip_pair> mfma_pairs;
    std::vector<uint64_t> mfma_off;
    uint32_t mfma_max_n1 = 0;
    std::vector<char> claimed(n_pairs, 0);
    if (use_mfma)
        for (uint32_t p = 0; p < n_pairs; p++)
        {
            const uint32_t na = ctx->img_n[pairs[p].image_1], nb = ctx->img_n[pairs[p].image_2];
            if (na == 0 || nb == 0 || nb > MFMA_MAX_REFS)
                continue;
            claimed[p] = 1;
            mfma_pairs.push_back(pairs[p]);
            mfma_off.push_back(out_offset[p]);
            mfma_max_n1 = std::max(mfma_max_n1, na);
            ctx->match_computed += (uint64_t)na * nb;
            ctx->match_delivered += (uint64_t)na * nb;
        }
    const uint32_t n_mfma = (uint32_t)mfma_pairs.size();
    // ---- of the others, pairs whose reverse is in the batch too are matched in both directions from one pass over their distances
    constexpr bool use_sym = true;
    std::vector<sym_job> sym;
    std::vector<ochip_pair> single_pairs;
    std::vector<uint64_t> single_off;
    uint64_t part_total = 0, part_max = 0;
    uint32_t sym_max_na = 0, sym_max_nb = 0;
    std::vector<uint32_t> sym_groups{0}; // first job of every group: a group's partials fit the cap, groups run one after the other
    {
        std::unordered_map<uint64_t, uint32_t> first; // (image_1, image_2) -> first pair with these images
        if (use_sym)
            for (uint32_t p = 0; p < n_pairs; p++)
                if (!claimed[p])
                    first.emplace(((uint64_t)pairs[p].image_1 << 32) | pairs[p].image_2, p);
        for (uint32_t p = 0; p < n_pairs; p++)
        {
            if (claimed[p])
                continue;
            const uint32_t a = pairs[p].image_1, b = pairs[p].image_2;
            const uint32_t na = ctx->img_n[a], nb = ctx->img_n[b];
            // (the partials of the paired jobs that are in flight together are capped at 8 GB per context - OCHIP_MATCH_SYM_CAP_MB
            // overrides it -: when the next pair would not fit, a new group starts, which re-uses the buffer after the
            // previous group's merge; a single pair larger than the cap goes one direction at a time)
            static const uint64_t cap_bytes = getenv("OCHIP_MATCH_SYM_CAP_MB") ? (uint64_t)atoll(getenv("OCHIP_MATCH_SYM_CAP_MB")) << 20 : (8ull << 30);
            const uint64_t need = (uint64_t)((na + 63) / 64) * nb;
            if (use_sym && a != b && na > 0 && nb > 0 && need * sizeof(uint2) <= cap_bytes)
            {
                auto it = first.find(((uint64_t)b << 32) | a);
                if (it != first.end() && it->second != p && !claimed[it->second])
                {
                    claimed[p] = claimed[it->second] = 1;
                    if ((part_total + need) * sizeof(uint2) > cap_bytes)
                    {
                        sym_groups.push_back((uint32_t)sym.size());
                        part_total = 0;
                    }
                    sym.push_back(sym_job{a, b, out_offset[p], out_offset[it->second], part_total});
                    part_total += need;
                    part_max = std::max(part_max, part_total);
                    ctx->match_computed += (uint64_t)na * nb;
                    ctx->match_delivered += 2 * (uint64_t)na * nb;
                    sym_max_na = std::max(sym_max_na, na);
                    sym_max_nb = std::max(sym_max_nb, nb);
                    continue;
                }
            }
            claimed[p] = 1;
            ctx->match_computed += (uint64_t)na * nb;
            ctx->match_delivered += (uint64_t)na * nb;
            single_pairs.push_back(pairs[p]);
            single_off.push_back(out_offset[p]);
        }
    }
    const uint32_t n_single = (uint32_t)single_pairs.size(), n_sym = (uint32_t)sym.size();
    uint32_t max_n1_single = 0;
    for (const ochip_pair &pr : single_pairs)
        max_n1_single = std::max(max_n1_single, ctx->img_n[pr.image_1]);

    if (ctx->img_tables_dirty)
    {
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->img_off_dev, ctx->img_off.data(), (size_t)ctx->n_images * 8,
                                      hipMemcpyHostToDevice, ctx->stream));
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->img_n_dev, ctx->img_n.data(), (size_t)ctx->n_images * 4,
                                      hipMemcpyHostToDevice, ctx->stream));
        ctx->img_tables_dirty = false;
    }
    if (n_single + n_mfma > ctx->pairs_cap)
    {
        if (ctx->pairs_dev)
            OCHIP_HIP(ctx, hipFree(ctx->pairs_dev));
        if (ctx->out_off_dev)
            OCHIP_HIP(ctx, hipFree(ctx->out_off_dev));
        ctx->pairs_dev = nullptr;
        ctx->out_off_dev = nullptr;
        ctx->pairs_cap = 0;
        if (hipMalloc((void **)&ctx->pairs_dev, (size_t)(n_single + n_mfma) * sizeof(ochip_pair)) != hipSuccess ||
            hipMalloc((void **)&ctx->out_off_dev, (size_t)(n_single + n_mfma) * 8) != hipSuccess)
            return ochip_fail(ctx, OCHIP_ENOMEM, "hipMalloc for the pair table failed");
        ctx->pairs_cap = n_single + n_mfma;
    }
    if (n_mfma)
    {
        // operands of the matrix-core kernel for the features uploaded since the last launch
        int rc = ochip_ensure(ctx, &ctx->desc_fp4_dev, &ctx->desc_fp4_cap, (size_t)ctx->desc_capacity * 256);
        if (rc == OCHIP_OK)
            rc = ochip_ensure(ctx, &ctx->desc_negpop_dev, &ctx->desc_negpop_cap, (size_t)ctx->desc_capacity * 4);
        if (rc == OCHIP_OK)
            rc = ochip_ensure(ctx, &ctx->desc_pop_dev, &ctx->desc_pop_cap, (size_t)ctx->desc_capacity * 4);
        if (rc)
            return rc;
        if (ctx->fp4_valid < ctx->desc_used)
        {
            const uint64_t n_new = ctx->desc_used - ctx->fp4_valid;
            hipLaunchKernelGGL(expand_fp4_kernel, dim3((uint32_t)((n_new + 3) / 4)), dim3(256), 0, ctx->stream, ctx->desc_dev,
                               ctx->fp4_valid, n_new, (uint32_t *)ctx->desc_fp4_dev, (float *)ctx->desc_negpop_dev,
                               (uint32_t *)ctx->desc_pop_dev);
            ctx->fp4_valid = ctx->desc_used;
        }
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->pairs_dev + n_single, mfma_pairs.data(), (size_t)n_mfma * sizeof(ochip_pair),
                                      hipMemcpyHostToDevice, ctx->stream));
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->out_off_dev + n_single, mfma_off.data(), (size_t)n_mfma * 8, hipMemcpyHostToDevice,
                                      ctx->stream));
    }
    {
        void *p = ctx->match_out_dev;
        size_t cap = ctx->match_out_cap * sizeof(ochip_match);
        int rc = ochip_ensure(ctx, &p, &cap, (size_t)(out_total ? out_total : 1) * sizeof(ochip_match));
        ctx->match_out_dev = (ochip_match *)p;
        ctx->match_out_cap = cap / sizeof(ochip_match);
        if (rc)
            return rc;
        if (n_sym)
        {
            rc = ochip_ensure(ctx, &ctx->sym_jobs_dev, &ctx->sym_jobs_cap, (size_t)n_sym * sizeof(sym_job));
            if (rc == OCHIP_OK)
                rc = ochip_ensure(ctx, &ctx->sym_part_dev, &ctx->sym_part_cap, (size_t)part_max * sizeof(uint2));
            if (rc)
                return rc;
        }
    }
    if (n_single)
    {
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->pairs_dev, single_pairs.data(), (size_t)n_single * sizeof(ochip_pair), hipMemcpyHostToDevice,
                                      ctx->stream));
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->out_off_dev, single_off.data(), (size_t)n_single * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    if (n_sym)
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->sym_jobs_dev, sym.data(), (size_t)n_sym * sizeof(sym_job), hipMemcpyHostToDevice, ctx->stream));
    // the pageable sources above must be consumed before they go out of scope
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    if (max_n1 == 0)
        return OCHIP_OK;

    constexpr int qpt = 1; // queries per lane of the popcount kernel (2 and 4 measured slower: register pressure)
    const uint32_t chunks = (max_n1_single + BLOCK * qpt - 1) / (BLOCK * qpt);
    const uint64_t blocks = (uint64_t)chunks * n_single;
    const uint32_t sym_chunks = (sym_max_na + BLOCK - 1) / BLOCK;
    const uint64_t sym_blocks = (uint64_t)sym_chunks * n_sym;
    const uint32_t mfma_chunks = (mfma_max_n1 + MFMA_QUERIES - 1) / MFMA_QUERIES;
    const uint64_t mfma_blocks = (uint64_t)mfma_chunks * n_mfma;
    if (blocks > 0x7FFFFFFFull || sym_blocks > 0x7FFFFFFFull || mfma_blocks > 0x7FFFFFFFull)
        return ochip_fail(ctx, OCHIP_EINVAL, "batch too large: %llu workgroups", (unsigned long long)(blocks + sym_blocks + mfma_blocks));
    hipEvent_t e0, e1;
    ochip_prof_begin(ctx, OCHIP_K_MATCH, &e0, &e1);
    if (mfma_blocks)
        hipLaunchKernelGGL(hamming_2nn_mfma_kernel, dim3((uint32_t)mfma_blocks), dim3(256), 0, ctx->stream,
                           (const uint4 *)ctx->desc_fp4_dev, (const float *)ctx->desc_negpop_dev,
                           (const uint32_t *)ctx->desc_pop_dev, ctx->img_off_dev, ctx->img_n_dev, ctx->pairs_dev + n_single,
                           ctx->out_off_dev + n_single, ctx->match_out_dev, mfma_chunks);
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((uint32_t)blocks), dim3(BLOCK), 0, ctx->stream, ctx->desc_dev,
                           ctx->img_off_dev, ctx->img_n_dev, ctx->pairs_dev, ctx->out_off_dev, ctx->match_out_dev,
                           chunks);
    };
    if (blocks)
        launch(hamming_2nn_kernel<1>);
    sym_groups.push_back(n_sym);
    for (size_t gi = 0; gi + 1 < sym_groups.size(); gi++)
    {
        const uint32_t g0 = sym_groups[gi], gn = sym_groups[gi + 1] - g0;
        if (gn == 0)
            continue;
        const sym_job *jobs = (const sym_job *)ctx->sym_jobs_dev + g0;
        hipLaunchKernelGGL(hamming_2nn_sym_kernel, dim3(sym_chunks * gn), dim3(BLOCK), 0, ctx->stream, ctx->desc_dev, ctx->img_off_dev,
                           ctx->img_n_dev, jobs, ctx->match_out_dev, (uint2 *)ctx->sym_part_dev, sym_chunks);
        hipLaunchKernelGGL(sym_merge_kernel, dim3(gn, (sym_max_nb + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, ctx->stream, jobs, ctx->img_n_dev,
                           (const uint2 *)ctx->sym_part_dev, ctx->match_out_dev);
    }
    ochip_prof_end(ctx, OCHIP_K_MATCH, e0, e1);
    OCHIP_HIP(ctx, hipGetLastError());
    return OCHIP_OK;
}

int ochip_match_fetch(ochip_ctx *ctx, ochip_match *out, uint64_t out_total)
{
    if (!ctx)
        return OCHIP_EINVAL;
    if (out_total != ctx->match_out_total)
        return ochip_fail(ctx, OCHIP_EINVAL, "out_total %llu differs from the launch (%llu)",
                          (unsigned long long)out_total, (unsigned long long)ctx->match_out_total);
    if (out_total == 0)
        return OCHIP_OK;
    if (!out)
        return ochip_fail(ctx, OCHIP_EINVAL, "out is NULL");
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    OCHIP_HIP(ctx, hipMemcpyAsync(out, ctx->match_out_dev, (size_t)out_total * sizeof(ochip_match),
                                  hipMemcpyDeviceToHost, ctx->stream));
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    return OCHIP_OK;
}

int ochip_match_batch(ochip_ctx *ctx, const ochip_pair *pairs, uint32_t n_pairs, const uint64_t *out_offset,
                      ochip_match *out)
{
    if (!ctx)
        return OCHIP_EINVAL;
    uint64_t total = 0;
    for (uint32_t p = 0; p < n_pairs; p++)
    {
        if (pairs[p].image_1 >= ctx->n_images)
            return ochip_fail(ctx, OCHIP_EINVAL, "pair %u: image out of range", p);
        const uint64_t end = out_offset[p] + ctx->img_n[pairs[p].image_1];
        total = end > total ? end : total;
    }
    int rc = ochip_match_launch(ctx, pairs, n_pairs, out_offset, total);
    if (rc)
        return rc;
    return ochip_match_fetch(ctx, out, total);
}

} // extern "C"
