// elbo_rows on the matrix cores: the per-observation part of one SVI step for 16 rows per wave.
//
// Reference arithmetic restated (gdrf/models/sparse_gdrf.py:354-372,396-409, abstract_gdrf.py:21-22):
//   v_kn = clamp(variance - q_n, 0) + tt_kn ; mu = loc + v eps (+ mean) ; theta = softmax_k(mu) ; p = theta Phi ;
//   log-likelihood sum_v w_v log clamp(p_v / sum p) ; both Normal sites ; row-local backward (vbar, locbar) ; Phi-bar.
//
// The one-thread-per-row kernel (kernels_n.h: elbo_rows_kernel) spends ~3000 LDS reads per row on the three small products
// p = theta Phi, thetabar = pbar Phi^T and Phi-bar += theta^T pbar (K x V each), holds 209 registers and runs two waves per SIMD:
// 0.67 ms at N = 1e6 for 0.45 GB of traffic.  Here the three products are 16x16x4 matrix instructions whose operands never leave the
// registers: lane (lr, lg) = (lane & 15, lane >> 4) of a wave owns row lr of the wave's 16 rows and, of every 16-wide block of topics or
// words, the four indices crow(lg, r) = the rows of the accumulator tile that the lane holds - so
//   P^T      = Phi^T Theta^T : B operand = the lane's own theta,  accumulator = p of the lane's own words,
//   Thetabar^T = Phi pbar^T  : B operand = the lane's own pbar,   accumulator = thetabar of the lane's own topics
// (the reduction slot of a matrix instruction may enumerate topics / words in any order as long as A and B agree), and only
// Phi-bar += Theta^T pbar, which reduces over the ROWS, passes theta and pbar through a wave-private LDS tile.  Row reductions (softmax
// maximum and sum, sum_v p, the pull-back's inner product) are two shuffles across the four lanes of a row; the counts are read as the
// 16-word runs the accumulator layout asks for, each byte once.  Bit-deterministic: no float atomics, fixed group -> wave map.
#pragma once
#include "common.h"
#include "kernels_mm.h"

namespace gdrf {

// float: the hardware transcendental / reciprocal instructions (v_exp_f32, v_log_f32, v_rcp_f32: ~1 ulp), a handful of instructions where
// the library calls are 20-40 with branches - this kernel evaluates 20 logarithms, 4 exponentials and ~30 quotients per row and lane.
// double keeps the library functions.
template <typename T> __device__ __forceinline__ T r_exp(T x) { return t_exp<T>(x); }
template <> __device__ __forceinline__ float r_exp<float>(float x) { return __expf(x); }
template <typename T> __device__ __forceinline__ T r_log(T x) { return t_log<T>(x); }
template <> __device__ __forceinline__ float r_log<float>(float x) { return __logf(x); }
template <typename T> __device__ __forceinline__ T r_rcp(T x) { return T(1) / x; }
template <> __device__ __forceinline__ float r_rcp<float>(float x) { return __builtin_amdgcn_rcpf(x); }

template <typename T> __device__ __forceinline__ T rows4_sum(T v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
template <typename T> __device__ __forceinline__ T rows4_max(T v) { v = fmax(v, __shfl_xor(v, 16, 64)); v = fmax(v, __shfl_xor(v, 32, 64)); return v; }

// LDS bytes: scratch 128 | phiS K V | per wave: thT 16 x (16 NKT + 1), pbT 16 x (16 NVT + 1) | per wave Phi-bar slab K V
template <typename T> static inline size_t rows_mfma_lds(int K, int V, int nkt, int nvt, int waves) {
  return 128 + ((size_t)K * V + (size_t)waves * (16 * (16 * nkt + 1) + 16 * (16 * nvt + 1)) + (size_t)waves * K * V) * sizeof(T);
}

// element at a 32-bit BYTE offset from a wave-uniform base: global_load with the base in scalar registers and one offset register
template <typename P> __device__ __forceinline__ P at_u32(const P* base, unsigned byte_off) {
  return *reinterpret_cast<const P*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename P> __device__ __forceinline__ void put_u32(P* base, unsigned byte_off, P v) {
  *reinterpret_cast<P*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// raw per-group operands of one lane, loaded one group ahead of their use
template <typename T, int NKT, int NVT> struct RowsRaw {
  T qn, ek[NKT][4], tt[NKT][4], lc[NKT][4];
  int32_t w[NVT][4];
};

template <typename T, int NKT, int NVT>
__global__ __launch_bounds__(256) void elbo_rows_mfma_kernel(
    int64_t nrows, int K, int V, const Hyper* __restrict__ h,
    const T* __restrict__ qpart, int nqpart, const T* __restrict__ loc, const T* __restrict__ tt, const T* __restrict__ eps,
    int64_t ldk, int64_t lde, const int32_t* __restrict__ ws, const T* __restrict__ phi,
    const T* __restrict__ mean /*may be null*/, int64_t mean_sk, int64_t mean_sn,
    T* __restrict__ qout, T* __restrict__ vbar, T* __restrict__ locbar, T* __restrict__ asum, T* __restrict__ mu_out,
    double* __restrict__ dpart /*[grid][4]*/, T* __restrict__ phibar_part /*[grid][K*V]*/) {
  using MF = Mfma<T>;
  using acc_t = typename MF::acc_t;
  using Raw = RowsRaw<T, NKT, NVT>;
  constexpr int TLK = 16 * NKT + 1, TLV = 16 * NVT + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* scratch = reinterpret_cast<double*>(smem);        // [16]
  T* phiS = reinterpret_cast<T*>(smem + 128);               // [K][V]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  T* thT = phiS + K * V + (size_t)wave * (16 * TLK + 16 * TLV);     // [16 rows][TLK]
  T* pbT = thT + 16 * TLK;                                           // [16 rows][TLV]
  T* slabs = phiS + K * V + (size_t)nwaves * (16 * TLK + 16 * TLV);  // [waves][K V]
  for (int e = threadIdx.x; e < K * V; e += blockDim.x) phiS[e] = phi[e];
  __syncthreads();
  const T var = (T)h->var, eta = (T)h->noise, feps = t_eps<T>();
  // fragments of Phi, re-read from LDS for every row group (32 conflict-free reads against ~1500 instructions of row work: keeping the
  // 32 - 128 values in registers instead costs a wave per SIMD):
  //   frag2(vt, kt, s) = Phi[16 kt + crow(lg, s)][16 vt + lr]   (A operand of P^T = Phi^T Theta^T)
  //   frag4(kt, vt, r) = Phi[16 kt + lr][16 vt + crow(lg, r)]   (A operand of Thetabar^T = Phi pbar^T)
  auto frag2 = [&](int vt, int kt, int s) -> T {
    const int k2 = 16 * kt + MF::crow(lane, s), v2 = 16 * vt + lr;
    return (k2 < K && v2 < V) ? phiS[k2 * V + v2] : T(0);
  };
  auto frag4 = [&](int kt, int vt, int r) -> T {
    const int k4 = 16 * kt + lr, v4 = 16 * vt + MF::crow(lane, r);
    return (k4 < K && v4 < V) ? phiS[k4 * V + v4] : T(0);
  };
  acc_t accPhi[NKT][NVT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int vt = 0; vt < NVT; ++vt) accPhi[kt][vt] = acc_t{0, 0, 0, 0};
  double s_site = 0, s_llw = 0, s_noise = 0, s_vd = 0;
  const int64_t ngroups = (nrows + 15) / 16, gstride = (int64_t)gridDim.x * nwaves;
  // every load of a group is unconditional, from clamped addresses (a load inside `if (row and topic valid)` costs its own branch and its
  // own wait: ~40 dependent round trips per group), and issued one group AHEAD: the next group's 30 values travel while this one computes
  // addressing: a wave-uniform base (the group's first row) in scalar registers + ONE 32-bit lane offset per own topic / word - the host
  // routes sizes whose offsets would not fit 32 bits to the one-thread-per-row kernel
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  unsigned otop[NKT][4], oeps[NKT][4], owrd[NVT][4];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * kt + MF::crow(lane, r), kc = k < K ? k : K - 1;
      otop[kt][r] = (unsigned)((int64_t)kc * ldk * sizeof(T)); oeps[kt][r] = (unsigned)((int64_t)kc * lde * sizeof(T));      // byte offsets
    }
#pragma unroll
  for (int vt = 0; vt < NVT; ++vt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int v = 16 * vt + MF::crow(lane, r); owrd[vt][r] = (unsigned)(v < V ? v : V - 1) * 4u; }
  auto load_group = [&](int64_t g, Raw& R) {
    const int64_t n0 = g * 16;                                        // wave-uniform
    const int64_t left = nrows - n0;
    const unsigned lrc = (unsigned)((int64_t)lr < left ? lr : left - 1);   // rows past the end re-read the last row (values unused)
    const unsigned lrb = lrc * (unsigned)sizeof(T);
    const T* qb = qpart + n0; const T* eb = eps + n0; const T* tb_ = tt + n0; const T* lb = loc + n0;
    const int32_t* wb = ws + n0 * V;
    T q = 0;
    for (int c = lg; c < nqpart; c += 4) q += at_u32(qb, (unsigned)((int64_t)c * ldk * sizeof(T)) + lrb);      // the four lanes of a row share the partial row norms of W
    R.qn = q;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        R.ek[kt][r] = at_u32(eb, oeps[kt][r] + lrb);
        R.tt[kt][r] = at_u32(tb_, otop[kt][r] + lrb);
        R.lc[kt][r] = at_u32(lb, otop[kt][r] + lrb);
      }
    const unsigned wrow = lrc * (unsigned)V * 4u;
#pragma unroll
    for (int vt = 0; vt < NVT; ++vt)
#pragma unroll
      for (int r = 0; r < 4; ++r) R.w[vt][r] = at_u32(wb, wrow + owrd[vt][r]);
  };
  Raw nxt;
  int64_t g = (int64_t)blockIdx.x * nwaves + wave_u;
  if (g < ngroups) load_group(g, nxt);
  for (; g < ngroups; g += gstride) {
    const Raw cur = nxt;
    if (g + gstride < ngroups) load_group(g + gstride, nxt);
    const int64_t n0 = g * 16, n = n0 + lr;
    const bool ok = n < nrows;
    // ---- variance, draw, softmax over the topics (own topics: k = 16 kt + crow(lg, r)); branch-free, dead lanes carry neutral values
    const T qn = rows4_sum(cur.qn);
    const T a = (var - qn > T(0)) ? T(1) : T(0), v0 = a * (var - qn);
    T vk[NKT][4], mk[NKT][4], th[NKT][4];
    T mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * kt + MF::crow(lane, r);
        const bool live = ok && k < K;
        vk[kt][r] = live ? v0 + cur.tt[kt][r] : T(1);
        mk[kt][r] = live ? cur.lc[kt][r] + vk[kt][r] * cur.ek[kt][r] : T(-3.0e38f);
      }
    if (mean) {                                            // f_loc + mean_function(xs): rare, one uniform branch per group
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * kt + MF::crow(lane, r);
          if (ok && k < K) mk[kt][r] += mean[(int64_t)k * mean_sk + n * mean_sn];
        }
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmax(mx, mk[kt][r]);
    mx = rows4_max(mx);
    T se = 0;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        th[kt][r] = r_exp<T>(fmax(mk[kt][r] - mx, T(-200)));       // dead lanes: exp(-200) = 0 in either precision's softmax sum
        const int k = 16 * kt + MF::crow(lane, r);
        th[kt][r] = (ok && k < K) ? th[kt][r] : T(0);
        se += th[kt][r];
      }
    se = rows4_sum(se);
    const T ise = r_rcp<T>(ok ? se : T(1));
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * kt + MF::crow(lane, r);
        th[kt][r] = ok ? th[kt][r] * ise : ((k < K) ? T(1) / (T)K : T(0));      // padding rows: a harmless uniform theta (their counts are 0)
      }
    // ---- P^T = Phi^T Theta^T: p of the own words v = 16 vt + crow(lg, r)
    acc_t accP[NVT];
#pragma unroll
    for (int vt = 0; vt < NVT; ++vt) {
      accP[vt] = acc_t{0, 0, 0, 0};
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int s = 0; s < 4; ++s) accP[vt] = MF::mma(frag2(vt, kt, s), th[kt][s], accP[vt]);
    }
    T ps = 0;
#pragma unroll
    for (int vt = 0; vt < NVT; ++vt)
#pragma unroll
      for (int r = 0; r < 4; ++r) ps += accP[vt][r];
    ps = rows4_sum(ps);
    const T ips = r_rcp<T>(ps);
    T llw = 0;
    T pbar[NVT][4];
#pragma unroll
    for (int vt = 0; vt < NVT; ++vt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int v = 16 * vt + MF::crow(lane, r);
        const bool live = ok && v < V;
        const T p = accP[vt][r];
        const T wv = live ? (T)cur.w[vt][r] : T(0);
        const T ph = p * ips;
        const bool inr = (ph > feps) && (ph < T(1) - feps);
        llw += wv * r_log<T>(fmin(fmax(ph, feps), T(1) - feps));       // padding words: p = 0 -> log(eps) times a zero count
        pbar[vt][r] = (inr && live) ? wv * r_rcp<T>(p) : T(0);
      }
    s_llw += (double)llw;
    // ---- Thetabar^T = Phi pbar^T: thetabar of the own topics
    acc_t tb[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      tb[kt] = acc_t{0, 0, 0, 0};
#pragma unroll
      for (int vt = 0; vt < NVT; ++vt)
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[kt] = MF::mma(frag4(kt, vt, r), pbar[vt][r], tb[kt]);
    }
    // ---- softmax pull-back without the cancellation of thetabar_k - sum_j theta_j thetabar_j (see elbo_rows_kernel): reference value =
    // thetabar of the dominant topic, found by the four lanes of the row
    T cref = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) cref = (mk[kt][r] == mx) ? fmax(cref, tb[kt][r]) : cref;
    cref = rows4_max(cref);
    T dot = 0;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dot += th[kt][r] * (cref - tb[kt][r]);
    dot = rows4_sum(dot);
    T site = 0, ng = 0, vsum = 0;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * kt + MF::crow(lane, r);
        const bool live = ok && k < K;
        const T mub = th[kt][r] * ((tb[kt][r] - cref) + dot);
        const T vv = vk[kt][r], ekr = cur.ek[kt][r], s = vv + eta, is = r_rcp<T>(s), rr = vv * is, e2 = ekr * ekr;
        const T st = r_log<T>(rr) - T(0.5) * e2 * rr * rr + T(0.5) * e2;              // -log s + log v = log(v / s)
        const T dcdv = -is + r_rcp<T>(vv) - e2 * rr * eta * is * is;
        const T ngk = -is + e2 * rr * rr * is;
        const T vb = mub * ekr + dcdv;
        site += live ? st : T(0); ng += live ? ngk : T(0); vsum += live ? vb : T(0);
        if (live) {
          const unsigned ob = otop[kt][r] + (unsigned)lr * (unsigned)sizeof(T);
          put_u32(vbar + n0, ob, vb);
          put_u32(locbar + n0, ob, mub);
          if (mu_out) put_u32(mu_out + n0, ob, mk[kt][r]);
        }
      }
    vsum = rows4_sum(vsum);
    s_site += (double)site; s_noise += (double)ng;
    if (ok && lg == 0) {
      const T vd = a * vsum;
      asum[n] = vd; qout[n] = qn;
      s_vd += (double)vd;
    }
    // ---- Phi-bar += Theta^T pbar (reduction over the 16 rows): theta and pbar through this wave's tiles, [row][topic] / [row][word]
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) thT[lr * TLK + 16 * kt + MF::crow(lane, r)] = th[kt][r];
#pragma unroll
    for (int vt = 0; vt < NVT; ++vt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pbT[lr * TLV + 16 * vt + MF::crow(lane, r)] = pbar[vt][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // LDS is in order per wave: the tiles are complete for every lane's reads
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int row = 4 * s + lg;                              // reduction slot lg of step s <-> row 4 s + lg
      T fa[NKT], fb[NVT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) fa[kt] = thT[row * TLK + 16 * kt + lr];
#pragma unroll
      for (int vt = 0; vt < NVT; ++vt) fb[vt] = pbT[row * TLV + 16 * vt + lr];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int vt = 0; vt < NVT; ++vt) accPhi[kt][vt] = MF::mma(fa[kt], fb[vt], accPhi[kt][vt]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // reads done before the next group's writes
  }
  // ---- Phi-bar of this workgroup: the waves' accumulators through their slabs, summed in wave order
  T* slab = slabs + (size_t)wave * K * V;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int vt = 0; vt < NVT; ++vt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * kt + MF::crow(lane, r), v = 16 * vt + lr;
        if (k < K && v < V) slab[k * V + v] = accPhi[kt][vt][r];
      }
  __syncthreads();
  for (int e = threadIdx.x; e < K * V; e += blockDim.x) {
    T s = 0;
    for (int w = 0; w < nwaves; ++w) s += slabs[(size_t)w * K * V + e];
    phibar_part[(int64_t)blockIdx.x * K * V + e] = s;
  }
  const double b0 = block_sum(s_site, scratch), b1 = block_sum(s_llw, scratch);
  const double b2 = block_sum(s_noise, scratch), b3 = block_sum(s_vd, scratch);
  if (threadIdx.x == 0) {
    dpart[4 * (int64_t)blockIdx.x + 0] = b0; dpart[4 * (int64_t)blockIdx.x + 1] = b1;
    dpart[4 * (int64_t)blockIdx.x + 2] = b2; dpart[4 * (int64_t)blockIdx.x + 3] = b3;
  }
}

}  // namespace gdrf
