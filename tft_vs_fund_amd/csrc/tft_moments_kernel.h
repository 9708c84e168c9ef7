// Normalize2Ddata x 3 and the 96 moment sums of linearTFT's design matrix as a kernel of their own: ONE triplet per wavefront, the
// triplet's correspondences read from HBM exactly once.
//
// Why (profiles/r4_headline_phases.txt): inside k_linear_tft_pose_rows the two normalisation passes were 31 % of a wavefront's cycles and
// memory-bound -- four triplets' correspondences (4 x 48 N bytes) cannot be staged in that kernel's LDS, so the second pass re-read them
// through the fabric (2.7x the algorithmic bytes at the L2 boundary), at two wavefronts per SIMD (19 KB of LDS and 256 registers each), the
// first pass a pure HBM wait in a synchronised first round, and all of it quantised to 2 500 wavefronts on 2 048 slots.  None of that is
// needed for THIS part of the path: it needs 48 accumulators, no workspace, and a triplet's 48 N bytes fit a wavefront's share of the LDS.
//   * one wavefront = one triplet (10 000 units on 1 024 SIMDs: no tail to speak of), three wavefronts per SIMD;
//   * pass 1 (centroids, Normalize2Ddata.m:33) loads every correspondence once, 16 bytes per lane and coalesced, and parks it in LDS;
//   * pass 2 (mean distances :35 and the moment sums) reads the LDS copy: two lanes share a correspondence exactly as in
//     tft_rows_kernel.h::rows_distances_moments (the even lane the 48 sums of {x1^2, x1 y1, x1}, the odd lane those of {y1^2, y1, 1}),
//     sums over the CENTRED coordinates, every moment multiplied by its power of the three scales afterwards;
//   * 48 -> 3 values per lane by a halving butterfly over the 32 lanes of equal parity (v_permlane32/16_swap, DPP), one plain step;
//   * out: 96 moments | 9 normalisation entries per triplet (PRE_DOUBLES = 112 doubles, 896 B against 9.6 KB read at N = 200).
// k_linear_tft_pose_rows<true> / k_gh_linear_rows<true> start from that record.  N beyond the LDS budget: the second pass re-reads global
// memory (STAGE = false).  Same arithmetic per correspondence as the row kernels; the sums are taken in a different order (32 pairs per trip
// instead of 8), so moments agree with theirs to rounding.
//
// Reference: auxiliar_functions/Normalize2Ddata.m:33-39, TFT_methods/linearTFT.m:36-62 (the rows of A whose Gram matrix these sums form).
#pragma once
#include "tft_kernel.h"

namespace tff {

constexpr int PRE_DOUBLES = 112;         // per triplet: mom[96] | nrm[9] | pad (16-byte aligned records)
constexpr int PRE_STAGE_MAX_N = 272;     // 48 N bytes <= 13 KB: twelve wavefronts per CU keep their triplets in LDS

struct MomentArgs {
    const double* corresp;   // B x (6 x N)
    long B;
    int N;
    double* pre;             // B x PRE_DOUBLES
};

inline size_t moments_lds_bytes(int N, bool stage) { return stage ? (size_t)6 * (size_t)N * sizeof(double) + 16 : 16; }
inline unsigned moments_grid(long B) { return (unsigned)(B > 0 ? (B < (1L << 20) ? B : (1L << 20)) : 1); }

template <bool STAGE>
__global__ void __launch_bounds__(64, 3) k_tft_moments(const MomentArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    const int lane = lane_id();
    const int slot = lane >> 1;
    const bool odd = (lane & 1) != 0;
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        const int N = opaque_int(a.N);
        if (N < 7) continue;                                                 // (wave-uniform; the pose kernel reports ST_TOO_FEW)
        const double* pts = a.corresp + b * 6 * (long)N;
        wave_sync();                                                         // (the previous triplet's LDS copy is no longer read)
        // ---- pass 1: centroids (Normalize2Ddata.m:33), and the triplet into LDS
        double c[6];
        {
            double sm[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 1
            for (int i0 = 0; i0 < N; i0 += 4 * WAVE) {                       // four trips' loads in flight
                Pt6 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = i0 + WAVE * u + lane; q[u] = load_pt(pts, (i < N) ? i : 0); }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + WAVE * u + lane;
                    const bool have = i < N;
#pragma unroll
                    for (int k = 0; k < 6; ++k) sm[k] += have ? q[u].v[k] : 0.0;
                    if (STAGE && have) {
                        double2* d = reinterpret_cast<double2*>(smem + 6 * i);
                        double2 t0, t1, t2;
                        t0.x = q[u].v[0]; t0.y = q[u].v[1]; t1.x = q[u].v[2]; t1.y = q[u].v[3]; t2.x = q[u].v[4]; t2.y = q[u].v[5];
                        d[0] = t0; d[1] = t1; d[2] = t2;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) c[k] = wave_sum(sm[k]) / (double)N;
        }
        wave_sync();
        const double* src = STAGE ? smem : pts;
        // ---- pass 2: mean distances (:35) + the 96 moment sums over the centred coordinates (tft_rows_kernel.h::rows_distances_moments)
        double acc[48];
#pragma unroll
        for (int k = 0; k < 48; ++k) acc[k] = 0.0;
        double dA = 0.0, dB = 0.0;
        auto body = [&](const Pt6& q, double& r2_out) {
            const double x1 = q.v[0] - c[0], y1 = q.v[1] - c[1];
            const double x2 = q.v[2] - c[2], y2 = q.v[3] - c[3];
            const double x3 = q.v[4] - c[4], y3 = q.v[5] - c[5];
            const double r1 = x1 * x1 + y1 * y1, r2 = x2 * x2 + y2 * y2, r3 = x3 * x3 + y3 * y3;
            dA += sqrt(odd ? r3 : r1);                                       // even lane: view 1, odd lane: view 3
            r2_out = r2;
            const double q2[4] = {1.0, x2, y2, r2};
            const double q3[4] = {1.0, x3, y3, r3};
            const double pa = odd ? y1 * y1 : x1 * x1, pb = odd ? y1 : x1 * y1, pc = odd ? 1.0 : x1;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const double wv = q3[bb] * q2[cc];
                    acc[4 * bb + cc] += pa * wv;
                    acc[16 + 4 * bb + cc] += pb * wv;
                    acc[32 + 4 * bb + cc] += pc * wv;
                }
        };
        constexpr int STEP = WAVE / 2;                                       // 32 correspondences per trip, two trips per loop iteration
        if constexpr (STAGE) {                                               // LDS copy: no prefetch across iterations (24 registers less; three wavefronts per SIMD cover the ds latency)
#pragma unroll 1
            for (int i = slot; i < N; i += 2 * STEP) {
                double r2a, r2b = 0.0;
                { const Pt6 q = load_pt(src, i); body(q, r2a); }
                if (i + STEP < N) { const Pt6 r = load_pt(src, i + STEP); body(r, r2b); }
                dB += sqrt(odd ? r2b : r2a);                                 // view 2: the even lane takes the first correspondence's root, the odd lane the second's
            }
        } else {
            Pt6 pe = load_pt(src, (slot < N) ? slot : 0);
            Pt6 po = load_pt(src, (slot + STEP < N) ? slot + STEP : 0);
#pragma unroll 1
            for (int i = slot; i < N; i += 2 * STEP) {
                const Pt6 q = pe;
                if (i + 2 * STEP < N) pe = load_pt(src, i + 2 * STEP);
                double r2a, r2b = 0.0;
                body(q, r2a);
                if (i + STEP < N) {
                    const Pt6 r = po;
                    if (i + 3 * STEP < N) po = load_pt(src, i + 3 * STEP);
                    body(r, r2b);
                }
                dB += sqrt(odd ? r2b : r2a);
            }
        }
        const double d1 = wave_sum(odd ? 0.0 : dA), d3 = wave_sum(odd ? dA : 0.0), d2 = wave_sum(dB);
        const double r2c = sqrt(2.0);
        const double dd[3] = {d1, d2, d3};
        double nr[9];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const double norm0 = dd[v] / (double)N;                          // :35
            nr[3 * v + 0] = r2c / norm0;                                     // :36
            nr[3 * v + 1] = -r2c * c[2 * v] / norm0;                         // :37
            nr[3 * v + 2] = -r2c * c[2 * v + 1] / norm0;
        }
        double* out = a.pre + b * PRE_DOUBLES;
        if (lane < 9) {
            double mine = nr[0];
#pragma unroll
            for (int k = 1; k < 9; ++k) mine = (lane == k) ? nr[k] : mine;
            out[96 + lane] = mine;
        }
        // sums over the 32 lanes of equal parity: halving butterfly (masks 32, 16, 8, 4), then one plain step (mask 2): 48 -> 3 values per lane
#pragma unroll
        for (int i = 0; i < 24; ++i) acc[i] = halve_sum<32>(acc[i], acc[i + 24]);
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = halve_sum<16>(acc[i], acc[i + 12]);
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i] = halve_sum<8>(acc[i], acc[i + 6]);
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = halve_sum<4>(acc[i], acc[i + 3]);
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] += dpp_mov<0x4E>(acc[i]);         // quad_perm [2,3,0,1]: lane ^ 2
        // the lane holds local indices base .. base + 2 of its parity's 48 sums; global moment index 48 * parity + local = 16 h + 4 i3 + i2
        const int base = 3 * ((lane >> 2) & 1) + 6 * ((lane >> 3) & 1) + 12 * ((lane >> 4) & 1) + 24 * ((lane >> 5) & 1);
        const double s1 = nr[0], s2 = nr[3], s3 = nr[6];
        if ((lane & 2) == 0) {                                               // (lanes l and l ^ 2 hold the same three sums)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int e = 48 * (lane & 1) + base + i;
                const int h = e >> 4, i3 = (e >> 2) & 3, i2 = e & 3;
                // degrees: p1 = {x^2, xy, x, y^2, y, 1} -> {2,2,1,2,1,0};  q = {1, x, y, x^2+y^2} -> {0,1,1,2}
                const double f1 = (h == 5) ? 1.0 : ((h == 2 || h == 4) ? s1 : s1 * s1);
                const double f3 = (i3 == 0) ? 1.0 : ((i3 == 3) ? s3 * s3 : s3);
                const double f2 = (i2 == 0) ? 1.0 : ((i2 == 3) ? s2 * s2 : s2);
                out[e] = acc[i] * (f1 * f3 * f2);
            }
        }
    }
}

// the row's record -> the LDS workspace of a row kernel (16 lanes: six 128-byte lines + the normalisation)
__device__ __forceinline__ void rows_load_pre(const double* pre, const long b, double* mom, double* nrm) {
    const int p = lane_id() & 15;
    const double* r = pre + b * PRE_DOUBLES;
#pragma unroll
    for (int k = 0; k < 6; ++k) mom[16 * k + p] = r[16 * k + p];
    if (p < 9) nrm[p] = r[96 + p];
}

}  // namespace tff
