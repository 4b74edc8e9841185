// kernels_separable.hpp -- the separable search: all translations of a rotation from one histogram pass over the points
//
// One of the kernel headers of libccal; each is included by exactly one translation unit
// (api_search.hip), so its __global__ functions have one definition in the library.
#pragma once

#include "kernels_common.hpp"

#pragma clang fp contract(off)

namespace ccal {

// ---------------------------------------------------------------------------
// Separable grid search
// ---------------------------------------------------------------------------
// On a translation x rotation product grid the folded residual separates:
//   |w.(p,1)| = |u_i(r) + b_t|,   u_i(r) = w123(r,j).(p_i - c_j) + a(r,j),   b_t = b(t,j)
// (w123 depends on the rotation only, the constant is a rotation part plus a
// translation part; both are taken relative to a per-cloud reference so that
// they stay small).  For one rotation and cloud, f(b) = sum_i |u_i + b| is a
// piecewise-linear function of the scalar b, so all translations are answered
// from ONE pass over the points: with the b_t sorted, point i falls into bin
// k_i = #{t : b_t < -u_i}; per bin a count and the sum of u; then with prefix
// sums N(q), S(q) over bins <= q
//   f(b_q) = 2 S(q) - S_total + b_q (2 N(q) - N).
// Work per (rotation, point) is two LDS reads and one LDS atomic instead of one
// residual per translation.  Sums are integers (u in 32-bit fixed point,
// accumulated with 64-bit integer atomics), so the result does not depend on the
// order of the atomics.
//
// One workgroup = RB rotations x one cloud x one range of at most TC translations.

struct SepMeta {
    float inv_delta;   // buckets per metre
    float offset;      // -(lowest threshold) * inv_delta
};

// Bucket of a key among n_buckets equal slices of the thresholds' range; monotone in the key
// (one correctly rounded fma, then a clamp).  Host (table construction) and device (lookup)
// evaluate this same float expression.
__host__ __device__ inline int separable_bucket(float key, const SepMeta& mt, int n_buckets)
{
    float fb = __builtin_fmaf(key, mt.inv_delta, mt.offset);
    const float top = (float)(n_buckets - 1);
#if defined(__HIP_DEVICE_COMPILE__)
    fb = __builtin_amdgcn_fmed3f(fb, 0.f, top);
#else
    fb = fb < 0.f ? 0.f : (fb > top ? top : fb);
#endif
    return (int)fb;
}

// Lookup-table entry (16 bits): four times the rank of the bucket's first threshold -- the byte offset
// of that threshold in the sorted list, and half the byte offset of its histogram word -- with the top
// bit set when the bucket holds two thresholds or more.  At most 8191 thresholds per launch.
constexpr unsigned kSepCrowded = 0x8000u, kSepOffsetMask = 0x7ffcu;

// BPB = buckets of the lookup table per histogram bin: with several the typical bucket holds no
// threshold or one, so the rank of a key is the table entry plus one or two comparisons.
//
// Fixed point: u * fx_scale rounded to a 32-bit integer (fx_scale = a power of two chosen by the
// caller so that |u| * fx_scale < 2^31 over the whole grid; the conversion saturates, so a wrong
// bound can only saturate values).  At least 2^-30 of the largest |u|: finer than the float u itself
// for all but the smallest residuals.
//
// A bin is ONE 64-bit word in LDS: count in the top 16 bits, the sum of the biased values (fixed point
// + 2^31, below 2^32) in the low 48, so a point costs one LDS atomic.  The word holds 65535 points
// without a carry into the count; before a workgroup has put that many points into its words (every 31
// trips of the point loop at 2048 points per trip) it empties its words into wide (32-bit count, 64-bit sum) accumulators in global memory
// -- its own slice, touched a handful of times per cloud, L2-resident.  With 8 bytes of LDS per bin a
// workgroup can serve RB = 4 rotations from one pass over the cloud's points (the pass is bound by the
// rate at which the points come out of L2: 16 bytes per point and workgroup).
//
// The point loop works on all 4 points x RB rotations of a trip at once, phase by phase (values and
// table reads; threshold reads; the rare crowded buckets; atomics), so that the LDS reads of a trip
// are in flight together instead of one round trip per item.
//
// NT = threads per workgroup: 512 where two or three workgroups share a CU, 1024 for the widest form,
// whose tables fill most of a CU's LDS.
template <int TC, int RB, int BPB, int NT>
__global__ __launch_bounds__(NT) void separable_hist_kernel(
    const f32x4* __restrict__ groups, const int* __restrict__ group_off,
    const f32x4* __restrict__ wr,            // [rotation][cloud]: w1, w2, w3, a
    const float* __restrict__ sb,            // [cloud][nt]: sorted translation constants
    const int* __restrict__ perm,            // [cloud][nt]: rank -> local translation index
    const unsigned short* __restrict__ tab,  // [cloud][M + 1]: see kSepCrowded; entry [M] = 4 nt
    const SepMeta* __restrict__ meta,        // [cloud]
    const double* __restrict__ scale,        // [cloud]: 1 / (n_clouds * n_points)
    const int64_t* __restrict__ cloud_pt_off,
    int n_clouds, int nt, int nr, float fx_scale,
    unsigned int* __restrict__ wide_cnt, long long* __restrict__ wide_sum,   // [workgroup][RB][TC + 1]
    double* __restrict__ partial /* [cloud][nt][nr] */)
{
    constexpr int M = BPB * TC;
    constexpr int kBins = RB * (TC + 1);
    __shared__ unsigned long long pk[RB][TC + 1];
    __shared__ float sbs[TC + 1];          // [nt] = +infinity
    __shared__ unsigned short tabs[M + 2];
    __shared__ unsigned long long tot_c[RB];
    __shared__ long long tot_s[RB], tot_all_s[RB];
    const int j = blockIdx.y;
    const int r0 = blockIdx.x * RB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kWaves = NT / 64;
    constexpr int kSepFlushTrips = 65535 / (kWaves * 256);   // a trip = 256 points per wave
    static_assert(4 * TC <= (int)kSepOffsetMask, "ranks must fit the lookup-table entry");
    const int64_t wg = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    unsigned int* wcnt = wide_cnt + wg * kBins;
    long long* wsum = wide_sum + wg * kBins;

    // thresholds, rotation vectors and the bucket function in units of the fixed point (fx_scale is a power
    // of two: the scaled values are the same numbers, and a residual converts to its integer as it is)
    for (int i = threadIdx.x; i <= nt; i += NT) sbs[i] = i < nt ? sb[(int64_t)j * nt + i] * fx_scale : __builtin_inff();
    for (int i = threadIdx.x; i <= M; i += NT) tabs[i] = tab[(int64_t)j * (M + 1) + i];
    for (int i = threadIdx.x; i < kBins; i += NT) {
        (&pk[0][0])[i] = 0ull;
        wcnt[i] = 0u;          // thread t owns bins t, t + NT, ... in every pass over them
        wsum[i] = 0ll;
    }
    if (threadIdx.x < RB) tot_all_s[threadIdx.x] = 0ll;
    // rotation vectors in pairs: the residuals of two rotations of one point are one packed FP32 chain
    static_assert(RB % 2 == 0, "rotations are handled in pairs");
    f32x2 w0[RB / 2], w1[RB / 2], w2[RB / 2], w3[RB / 2];
#pragma unroll
    for (int rp = 0; rp < RB / 2; ++rp) {
        const int ra = r0 + 2 * rp < nr ? r0 + 2 * rp : nr - 1;       // a partial last group repeats its last rotation
        const int rc = r0 + 2 * rp + 1 < nr ? r0 + 2 * rp + 1 : nr - 1;
        const f32x4 a = wr[(int64_t)ra * n_clouds + j] * fx_scale, c = wr[(int64_t)rc * n_clouds + j] * fx_scale;
        w0[rp] = f32x2{a[0], c[0]}; w1[rp] = f32x2{a[1], c[1]}; w2[rp] = f32x2{a[2], c[2]}; w3[rp] = f32x2{a[3], c[3]};
    }
    SepMeta mt = meta[j];
    mt.inv_delta /= fx_scale;
    // The two outer bins (every point beyond the thresholds: most points, for rotations far from the
    // optimum) never touch LDS.  Their counts are wave-wide population counts (scalar registers); of the
    // sums a lane keeps the one of ALL its points and the one of the points below the first threshold --
    // the sum above the last threshold is what is left once the bins in between are known.
    unsigned c_lo[RB] = {};
    unsigned long long s_all[RB] = {}, s_lo[RB] = {};
    const unsigned top8 = 8u * (unsigned)nt;
    __syncthreads();

    auto flush = [&]() {
        // every bin by the thread that owns it: plain read-modify-write, no atomics
        for (int i = threadIdx.x; i < kBins; i += NT) {
            const unsigned long long p = (&pk[0][0])[i];
            if (p == 0ull) continue;
            const unsigned long long c = p >> 48;
            wcnt[i] += (unsigned int)c;
            wsum[i] += (long long)(p & 0xffffffffffffull) - (long long)(c << 31);
            (&pk[0][0])[i] = 0ull;
        }
    };

    // One trip of a lane: the four points of its column of group g, all RB rotations, one point (RB items)
    // at a time and phase by phase.  Every item reads the table and its bucket's first threshold (an empty
    // bucket's entry names the NEXT threshold, which lies in a higher bucket and so above the key: the
    // comparison adds nothing; one past the last threshold sits +infinity).  All offsets stay in bytes: the
    // table entry is the offset of the threshold, twice that the offset of the histogram word.
    // kCheck: the group may hold padding points.
    auto trip = [&](const f32x4* gp, auto check) {
        constexpr bool kCheck = decltype(check)::value;
        const f32x4 x4 = gp[0], y4 = gp[16], z4 = gp[32];
        f32x4 o4 = {1.f, 1.f, 1.f, 1.f};
        if (kCheck) o4 = gp[48];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v[RB], thr[RB];
            unsigned e[RB];
            // phase 1: residuals of rotation pairs (packed FP32, three roundings), table read
            const f32x2 xx = {x4[u], x4[u]}, yy = {y4[u], y4[u]}, zz = {z4[u], z4[u]};
#pragma unroll
            for (int rp = 0; rp < RB / 2; ++rp) {
                const f32x2 v2 = __builtin_elementwise_fma(w0[rp], xx, __builtin_elementwise_fma(w1[rp], yy,
                                 __builtin_elementwise_fma(w2[rp], zz, w3[rp])));
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int rb = 2 * rp + h;
                    v[rb] = v2[h];
                    // k = #{q : sbs[q] < -v}.  The bucket function is monotone and the table was built with
                    // this very expression: every threshold of a lower bucket is below the key, every threshold
                    // of a higher bucket above it, so k is the table entry (thresholds in lower buckets) plus
                    // the thresholds of the key's own bucket that lie below it.
                    e[rb] = tabs[separable_bucket(-v[rb], mt, M)];
                }
            }
            // phase 2: the first threshold that could still lie below the key
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                thr[rb] = *(const float*)((const char*)sbs + (e[rb] & kSepOffsetMask));
            unsigned k8[RB];
            unsigned long long crowded = 0ull;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const bool below = thr[rb] < -v[rb];
                k8[rb] = 2u * (e[rb] & kSepOffsetMask) + (below ? 8u : 0u);
                crowded |= __builtin_amdgcn_ballot_w64(below) & __builtin_amdgcn_ballot_w64(e[rb] >= kSepCrowded);
            }
            // phase 3 (rare): a bucket with several thresholds, the first of them below the key
            if (crowded != 0ull) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    if (!(thr[rb] < -v[rb] && e[rb] >= kSepCrowded)) continue;
                    unsigned hi = (tabs[separable_bucket(-v[rb], mt, M) + 1] & kSepOffsetMask) >> 2;
                    unsigned kk = k8[rb] >> 3;
                    while (kk < hi) {
                        const unsigned mid = (kk + hi) >> 1;
                        if (sbs[mid] < -v[rb]) kk = mid + 1; else hi = mid;
                    }
                    k8[rb] = kk << 3;
                }
            }
            // phase 4: outer bins in registers, the others in the histogram
            const bool valid = !kCheck || o4[u] != 0.f;                      // not a padding point
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const unsigned biased = (unsigned)(int)v[rb] ^ 0x80000000u;   // towards zero: below one unit
                const bool lo = valid && k8[rb] == 0u, hi = k8[rb] == top8;
                c_lo[rb] += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(lo));
                s_all[rb] += valid ? biased : 0u;
                if (lo) s_lo[rb] += biased;
                if (valid && !lo && !hi)
                    atomicAdd((unsigned long long*)((char*)&pk[rb][0] + k8[rb]), (1ull << 48) + (unsigned long long)biased);
            }
        }
    };

    // points: a wave takes four 64-point groups per trip, a lane the four points of one column; padding
    // points exist in a cloud's last group only
    const int g0 = group_off[j], g1 = group_off[j + 1];
    const int64_t n_points = cloud_pt_off[j + 1] - cloud_pt_off[j];
    const int col = lane & 15, gsel = lane >> 4;
    int trips = 0;
    for (int gb = g0; gb < g1; gb += kWaves * 4) {
        const int g = gb + wave * 4 + gsel;
        if (g < g1) {
            const f32x4* gp = groups + (int64_t)g * 64 + col;
            if (__builtin_amdgcn_ballot_w64(g == g1 - 1) != 0ull) trip(gp, std::true_type{});
            else trip(gp, std::false_type{});
        }
        if (++trips == kSepFlushTrips) {          // uniform over the workgroup
            trips = 0;
            __syncthreads();
            flush();
            __syncthreads();
        }
    }
    __syncthreads();
    flush();
    __syncthreads();
    // the outer bins: the count below the first threshold is already a wave total; the sums are reduced
    // over the wave, and one lane per wave adds (c_lo, s_lo) to bin 0 and s_all to the rotation's total
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        long long sl = (long long)s_lo[rb], sa = (long long)s_all[rb];
        for (int off = 32; off > 0; off >>= 1) {
            sl += __shfl_xor(sl, off, 64);
            sa += __shfl_xor(sa, off, 64);
        }
        if (lane == 0) {
            atomicAdd(&wcnt[rb * (TC + 1)], c_lo[rb]);
            atomicAdd((unsigned long long*)&wsum[rb * (TC + 1)], (unsigned long long)(sl - ((long long)c_lo[rb] << 31)));
            atomicAdd((unsigned long long*)&tot_all_s[rb], (unsigned long long)sa);
        }
    }
    __threadfence_block();
    __syncthreads();

    // inclusive prefix sums over the bins in place, one wave per rotation; the totals go to LDS
    for (int rb = wave; rb < RB; rb += kWaves) {
        unsigned long long cn = 0;
        long long cs = 0;
        for (int base = 0; base <= nt; base += 64) {
            const int kk = base + lane;
            unsigned long long c = 0;
            long long sv = 0;
            if (kk <= nt) {
                c = wcnt[rb * (TC + 1) + kk];
                sv = wsum[rb * (TC + 1) + kk];
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned long long c2 = __shfl_up(c, off, 64);
                const long long s2 = __shfl_up(sv, off, 64);
                if (lane >= off) { c += c2; sv += s2; }
            }
            c += cn;
            sv += cs;
            if (kk <= nt) {
                wcnt[rb * (TC + 1) + kk] = (unsigned int)c;
                wsum[rb * (TC + 1) + kk] = sv;
            }
            cn = __shfl(c, 63, 64);
            cs = __shfl(sv, 63, 64);
        }
        // the top bin is never stored: the count of all points is the cloud's size, and the sum of
        // everything is s_all less the bias of all points
        if (lane == 0) {
            const unsigned long long n_all = (unsigned long long)n_points;
            tot_c[rb] = n_all;
            tot_s[rb] = tot_all_s[rb] - (long long)(n_all << 31);
        }
    }
    __threadfence_block();
    __syncthreads();

    const double sc = scale[j];
    const double inv_fx = 1.0 / (double)fx_scale;
    for (int q = threadIdx.x; q < nt; q += NT) {
        const double bq = (double)sbs[q] * inv_fx;
        const int tl = perm[(int64_t)j * nt + q];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if (r0 + rb >= nr) break;
            const double n_le = (double)wcnt[rb * (TC + 1) + q], n_all = (double)tot_c[rb];
            const double s_le = (double)wsum[rb * (TC + 1) + q], s_all = (double)tot_s[rb];
            const double f = (2.0 * s_le - s_all) * inv_fx + bq * (2.0 * n_le - n_all);
            partial[((int64_t)j * nt + tl) * nr + (r0 + rb)] = f * sc;
        }
    }
}

// cost[(t_off + tl) * row + r_off + rl] = sum over clouds (in cloud order) of partial[cloud][tl][rl]
__global__ __launch_bounds__(256) void separable_reduce_kernel(const double* __restrict__ partial,
                                                                int n_clouds, int nt, int nr,
                                                                float* __restrict__ cost, int64_t row,
                                                                int64_t t_off, int64_t r_off)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nt * nr) return;
    double s = 0.0;
    for (int j = 0; j < n_clouds; ++j) s += partial[(int64_t)j * nt * nr + i];
    const int64_t tl = i / nr, rl = i % nr;
    cost[(t_off + tl) * row + r_off + rl] = (float)s;
}


} // namespace ccal
