// NT GEMM core on f32/f64 MFMA 16x16x4:  C[m][n] = sum_k A[m][k] * Bt[n][k]
// Both operands have the reduction index contiguous in memory.  128x128 output tile per
// 256-thread workgroup (4 waves as 2x2, each wave 4x4 MFMA tiles), 128 bytes of k per row
// per chunk staged through LDS with a register prefetch of the next chunk.
//
// The reduction is a sequence of chunks c = 0 .. nchunks-1.  Chunk c uses A columns
// [kb + (c / R) * BK, +BK) and B "repeat" r = c % R of the same columns, where R = a_reuse():
// with R > 1 one staged A chunk serves R consecutive B chunks (each with its own per-row scale
// of A, read from an LDS table) -- the topic loop of the Wbar contraction, which would otherwise
// re-stream the W tile from HBM once per topic.
//
// The problem object P supplies operands and epilogue:
//   static constexpr bool SCALE_A         per-(repeat,row) scale of the A fragments from LDS
//   static constexpr bool A_PER_REP       the A chunk itself depends on the repeat (topic): staged for every chunk
//   static constexpr int  DEPTH           register prefetch distance in chunks (1, or 2 for cold HBM streams)
//   int  col_tiles(); bool loop_cols();   loop_cols: one workgroup walks every column tile of its row tile
//   int  a_reuse()
//   int  extra_lds_bytes()                problem-owned LDS behind the two staging tiles
//   void krange(m0, n0, bz, kb, ke)       A-column range [kb, ke), multiples of BK (triangular skipping)
//   void prepA(actx, m0, bz, extra) ; AVec loadA(actx, i, k, bz)  i = 0..VPT-1: the thread's i-th staged row
//   V    a_to_lds(AVec, actx, i, k)       conversion applied when the prefetched registers are written to LDS (so that a
//                                         precision change does not force a wait on the loads ahead of the MFMAs)
//   V    loadB(n0, i, k, rep, bz)
//   int  extra_chunks(); void fill_extra(As, Bs, x, m0, n0)   problem-staged chunks multiplied after the main reduction
//   void prepE(ectx, m0, bz)
//   void tile_done(acc, m0, n0, bz, ectx, wr, wc, lane)
//   void finish(m0, bz, ectx, smem, wr, wc, lane)     once per workgroup
#pragma once
#include "common.h"

namespace gdrf {

template <typename T> struct NTCfg {
  static constexpr int BK = GDRF_KBYTES_F64 > 0 && sizeof(T) == 8 ? GDRF_KBYTES_F64 / 8 : GDRF_KBYTES / (int)sizeof(T);   // 32 (f32) / 16 or 32 (f64)
  static constexpr int VE = 16 / (int)sizeof(T);            // elements per 16-byte vector
  static constexpr int LDK = BK + VE;                       // padded LDS row: 144 bytes
  static constexpr int KG = BK / 4;                         // k indices per lane group per chunk
  static constexpr int VPR = BK / VE;                       // 8 vectors per staged row
  static constexpr int VPT = GDRF_TILE * VPR / 256;         // 4 vectors per thread for the A tile (128 rows)
  // column-tile width: 128 for f32; 64 for f64, whose accumulators are twice as wide -- the narrower tile keeps a
  // workgroup at 64 accumulator registers per lane so that 2-3 workgroups share a CU and hide each other's
  // staging / epilogue phases (at 128 the f64 kernels fit one wave per SIMD and ran at ~35-58 % of the f64 MFMA peak)
  static constexpr int CW = sizeof(T) == 4 ? 128 : 64;
  static constexpr int NB = CW / 32;                        // 16-wide MFMA column tiles per wave
  static constexpr int VPTB = CW * VPR / 256;               // vectors per thread for the B tile (CW rows)
  static constexpr int LDS_BYTES = (GDRF_TILE + CW) * LDK * (int)sizeof(T);
};

// row (0..127) and k offset (elements) of the i-th vector a thread stages
template <typename T> __device__ __forceinline__ int nt_stage_row(int i) { return (int)(threadIdx.x / NTCfg<T>::VPR) + (256 / NTCfg<T>::VPR) * i; }
template <typename T> __device__ __forceinline__ int nt_stage_k() { return (int)(threadIdx.x % NTCfg<T>::VPR) * NTCfg<T>::VE; }

// triangular reductions at MFMA-tile granularity (a problem may declare `static constexpr int TRI`): krange() cuts the reduction at the
// granularity of a column tile (64 or 128 columns), so the chunks that cross the diagonal block multiply zeros for part of the 16-column
// MFMA tiles.  TRI = 1 (Bt[col][k] = 0 for k > col, the forward solve): a chunk starting at k is needed by the 16 columns from c0 only if
// k <= c0 + 15;  TRI = 2 (Bt[col][k] = 0 for k < col, the backward solve): only if k + BK - 1 >= c0.  The test is wave-uniform.
template <class P, class = void> struct NTTri { static constexpr int value = 0; };
template <class P> struct NTTri<P, decltype((void)P::TRI)> { static constexpr int value = P::TRI; };

// workgroups per CU the register allocation must leave room for (a problem may declare `static constexpr int MIN_WGS`)
template <class P, class = void> struct NTMinWgs { static constexpr int value = 2; };
template <class P> struct NTMinWgs<P, decltype((void)P::MIN_WGS)> { static constexpr int value = P::MIN_WGS; };

template <typename T, class P> __device__ __forceinline__ void gemm_nt_body(P& p);

#ifdef GDRF_NT_TRACE   // diagnostic builds only: per-workgroup phase stamps of the problems that declare TRACE (tools/nt_trace.py)
__device__ unsigned long long* g_nt_trace = nullptr;       // [workgroup][8]: block | column tile << 32, hw id, xcc id, t start, t first chunk staged, t loop end, t tile end, t end
template <class P, class = void> struct NTTrace { static constexpr bool value = false; };
template <class P> struct NTTrace<P, decltype((void)P::TRACE)> { static constexpr bool value = P::TRACE; };
#define NT_STAMP(slot) do { if (NTTrace<P>::value && g_nt_trace && threadIdx.x == 0 && blockIdx.y == 0) g_nt_trace[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define NT_STAMP(slot) do {} while (0)
#endif

template <typename T, class P>
__global__ __launch_bounds__(256, NTMinWgs<P>::value) void gemm_nt_kernel(P p) { gemm_nt_body<T, P>(p); }

// the same kernel capped at 160 registers: two of its waves and one 192-register wave (the bf16 TN contraction) fit a SIMD's
// 512 registers together.  (amdgpu_num_vgpr takes a literal only, hence a second entry point.)
template <typename T, class P>
__global__ __launch_bounds__(256, 3) __attribute__((amdgpu_num_vgpr(80))) void gemm_nt_kernel_v160(P p) { gemm_nt_body<T, P>(p); }

template <typename T, class P>
__device__ __forceinline__ void gemm_nt_body(P& p) {
  using C = NTCfg<T>;
  using V = typename Vec16<T>::type;
  using MM = Mfma<T>;
  using acc_t = typename MM::acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* As = reinterpret_cast<T*>(smem);
  T* Bs = As + GDRF_TILE * C::LDK;
  char* extra = smem + C::LDS_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lg = lane >> 4;
  const int wc_u = __builtin_amdgcn_readfirstlane(wc);
  const int nct = p.col_tiles();
  const bool loopc = p.loop_cols();
  int64_t rtile; int ct_first;
  p.map_block(blockIdx.x, nct, loopc, rtile, ct_first);
  const int ct_last = loopc ? nct : ct_first + 1;
  const int64_t m0 = rtile * GDRF_TILE;
  const int bz = p.batch_index(blockIdx.x, blockIdx.y);
  const int R = p.a_reuse();

#ifdef GDRF_NT_TRACE
  if (NTTrace<P>::value && g_nt_trace && threadIdx.x == 0 && blockIdx.y == 0) {
    g_nt_trace[(size_t)blockIdx.x * 8 + 0] = blockIdx.x | ((unsigned long long)ct_first << 32);
    g_nt_trace[(size_t)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID: wave, SIMD, CU, SH, SE
    g_nt_trace[(size_t)blockIdx.x * 8 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
  }
  NT_STAMP(3);
#endif
  typename P::ACtx actx;
  p.prepA(actx, m0, bz, extra);
  typename P::ECtx ectx;
  p.prepE(ectx, m0, bz);
  const T* scaleS = reinterpret_cast<const T*>(extra);     // [R][128] when SCALE_A

  const int srow_k = nt_stage_k<T>();
  for (int ct = ct_first; ct < ct_last; ++ct) {
    const int n0 = ct * C::CW;
    acc_t acc[4][C::NB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < C::NB; ++b) acc[a][b] = acc_t{0, 0, 0, 0};
    int kb, ke;
    p.krange(m0, n0, bz, kb, ke);
    const int nchunks = (ke > kb) ? ((ke - kb) / C::BK) * R : 0;
    // chunk c -> (A column kA, repeat rep).  With a shared A chunk the repeats are innermost (A staged once per R
    // chunks); when A itself depends on the repeat, k is innermost so each repeat's A stream is contiguous in HBM.
    const int nk = (ke > kb) ? (ke - kb) / C::BK : 0;
    auto chunk_pos = [&](int c, int& kA, int& rep) {
      if (P::A_PER_REP) { rep = c / nk; kA = kb + (c - rep * nk) * C::BK; }
      else { const int q = c / R; rep = c - q * R; kA = kb + q * C::BK; }
    };
    // register prefetch: DEPTH chunks ahead of the one being multiplied (DEPTH = 2 for operands that come cold
    // from HBM on every chunk; the sets are named so that every register index stays static)
    typename P::AVec ra0[C::VPT], ra1[C::VPT];
    V rb0[C::VPTB], rb1[C::VPTB];
    auto gload = [&](typename P::AVec (&ra)[C::VPT], V (&rb)[C::VPTB], int c) {
      int kA, rep;
      chunk_pos(c, kA, rep);
      if (rep == 0 || P::A_PER_REP) {
#pragma unroll
        for (int i = 0; i < C::VPT; ++i) ra[i] = p.loadA(actx, i, kA + srow_k, rep, bz);
      }
#pragma unroll
      for (int i = 0; i < C::VPTB; ++i) rb[i] = p.loadB(n0, i, kA + srow_k, rep, bz);
    };
    // multiply the staged chunk out of LDS (rep < 0: no per-row scale)
    auto compute = [&](int rep, int kA = 0) {
      T sc[4];
      bool need[C::NB];
#pragma unroll
      for (int b = 0; b < C::NB; ++b) {
        const int c0 = n0 + wc_u * (C::CW / 2) + b * 16;
        need[b] = NTTri<P>::value == 1 ? (kA <= c0 + 15) : (NTTri<P>::value == 2 ? (kA + C::BK - 1 >= c0) : true);
      }
      if (P::SCALE_A) {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) sc[t4] = rep >= 0 ? scaleS[rep * GDRF_TILE + wr * 64 + t4 * 16 + lr] : T(1);
      }
      // fragments: lane (lr, lg) owns k indices lg*KG .. lg*KG+KG-1 of this chunk for row/col lr
      const T* pa = &As[(wr * 64 + lr) * C::LDK + lg * C::KG];
      const T* pb = &Bs[(wc * (C::CW / 2) + lr) * C::LDK + lg * C::KG];
#pragma unroll
      for (int v = 0; v < C::KG / C::VE; ++v) {
        V fa[4], fb[C::NB];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          fa[t4] = *reinterpret_cast<const V*>(pa + t4 * 16 * C::LDK + v * C::VE);
          if (P::SCALE_A) {
#pragma unroll
            for (int e = 0; e < C::VE; ++e) fa[t4][e] *= sc[t4];
          }
        }
#pragma unroll
        for (int t4 = 0; t4 < C::NB; ++t4) fb[t4] = *reinterpret_cast<const V*>(pb + t4 * 16 * C::LDK + v * C::VE);
#pragma unroll
        for (int e = 0; e < C::VE; ++e)
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < C::NB; ++b) {
              if (NTTri<P>::value == 0 || need[b]) acc[a][b] = MM::mma(fa[a][e], fb[b][e], acc[a][b]);
            }
      }
    };
    auto body = [&](typename P::AVec (&ra)[C::VPT], V (&rb)[C::VPTB], int c) {
      int kA, rep;
      chunk_pos(c, kA, rep);
      __syncthreads();
      if (rep == 0 || P::A_PER_REP) {
#pragma unroll
        for (int i = 0; i < C::VPT; ++i) *reinterpret_cast<V*>(&As[nt_stage_row<T>(i) * C::LDK + srow_k]) = p.a_to_lds(ra[i], actx, i, kA + srow_k);
      }
#pragma unroll
      for (int i = 0; i < C::VPTB; ++i) *reinterpret_cast<V*>(&Bs[nt_stage_row<T>(i) * C::LDK + srow_k]) = rb[i];
      __syncthreads();
      if (c == 0) NT_STAMP(4);
      if (c + P::DEPTH < nchunks) gload(ra, rb, c + P::DEPTH);      // refill the set just consumed
      compute(rep, kA);
    };
    if (P::DEPTH == 1) {
      if (nchunks > 0) gload(ra0, rb0, 0);
      for (int c = 0; c < nchunks; ++c) body(ra0, rb0, c);
    } else {
      if (nchunks > 0) gload(ra0, rb0, 0);
      if (nchunks > 1) gload(ra1, rb1, 1);
      for (int c = 0; c < nchunks; c += 2) {
        body(ra0, rb0, c);
        if (c + 1 < nchunks) body(ra1, rb1, c + 1);
      }
    }
    // problem-owned extra chunks (a low-rank term of the epilogue done on the matrix cores): the problem writes
    // the two LDS tiles itself
    for (int x = 0; x < p.extra_chunks(); ++x) {
      __syncthreads();
      p.fill_extra(As, Bs, x, m0, n0);
      __syncthreads();
      compute(-1);
    }
    NT_STAMP(5);
    p.tile_done(acc, m0, n0, bz, ectx, wr, wc, lane);
    NT_STAMP(6);
  }
  p.finish(m0, bz, ectx, smem, wr, wc, lane);
  NT_STAMP(7);
}

// element (row, col) of accumulator register r of MFMA tile (a, b) inside the workgroup tile
template <typename T> __device__ __forceinline__ int nt_acc_row(int wr, int a, int lane, int r) {
  return wr * 64 + a * 16 + Mfma<T>::crow(lane, r);
}
template <typename T> __device__ __forceinline__ int nt_acc_col(int wc, int b, int lane) {
  return wc * (NTCfg<T>::CW / 2) + b * 16 + (lane & 15);
}

// default block -> (row tile, column tile) map: column tile fastest
// default staging: the prefetched A registers already hold the LDS representation
template <typename T> struct NTPlainA {
  using AVec = typename Vec16<T>::type;
  template <class... X> __device__ __forceinline__ AVec a_to_lds(const AVec& v, X&&...) const { return v; }
};
// default: no problem-owned extra chunks
struct NTNoExtra {
  __device__ __forceinline__ int extra_chunks() const { return 0; }
  template <typename T> __device__ __forceinline__ void fill_extra(T*, T*, int, int64_t, int) const {}
};

struct NTDefaultMap {
  __device__ __forceinline__ int batch_index(unsigned, unsigned by) const { return (int)by; }
  __device__ __forceinline__ void map_block(unsigned bid, int nct, bool loopc, int64_t& rtile, int& ct) const {
    if (loopc) { rtile = bid; ct = 0; } else { rtile = bid / nct; ct = (int)(bid % nct); }
  }
};
// XCD-aware map (speed only: blocks b and b+8 are observed to share an XCD, MI355X_MICROARCH.md): every
// XCD works on ONE column tile at a time so that tile's B panels stay resident in its 4 MiB L2, and
// the row tiles are dealt round-robin.  Bijective whenever nct divides 8; falls back otherwise.
struct NTXcdMap {
  __device__ __forceinline__ int batch_index(unsigned, unsigned by) const { return (int)by; }
  __device__ __forceinline__ void map_block(unsigned bid, int nct, bool loopc, int64_t& rtile, int& ct) const {
    if (loopc) { rtile = bid; ct = 0; return; }
    if (nct > 8 || (8 % nct) != 0 || (gridDim.x % 8u) != 0) { rtile = bid / nct; ct = (int)(bid % nct); return; }
    const unsigned xcd = bid & 7u, idx = bid >> 3, per = 8u / (unsigned)nct;   // XCDs per column tile
    ct = (int)(xcd % (unsigned)nct);
    rtile = (int64_t)idx * per + xcd / (unsigned)nct;
  }
};

// XCD-aware map that keeps ALL column tiles of a row tile on one XCD, dispatched back to back: their A operand (the
// row tile's slab, cold in HBM) is then fetched once into that XCD's L2 and shared, instead of once per column tile.
// Grid: nt_xcd_row_grid(rtiles, nct); padding blocks get rtile >= rtiles.
//
// The ORDER of the column tiles inside an XCD's sequence matters for triangular reductions (column tile ct costing ct + 1 or nct - ct
// units).  Measured with per-workgroup stamps (tools/nt_trace.py, profiles/r03/README.md): workgroup b of a launch runs on XCD b mod 8
// and, inside it, on shader engine (b / 8) mod 4 - a STATIC round robin - and workgroups are placed in order, so the engine that is
// handed the most work is always full (24 workgroups on its 8 CUs) while the launch waits for it and the three others starve.  In
// plain order (ct = 0, 1, 2, ...) engine e of every XCD only ever sees the column tiles ct = e mod 4 - 6 : 8 : 10 : 12 units - and
// the f64 solve GEMMs held 1.87 of the 3 workgroups per CU their registers allow.  The tiles are therefore dealt in a serpentine over
// groups of four (0 1 2 3 | 7 6 5 4 | 8 9 10 11 | 15 ...), the group counter running on across row tiles: every engine then sees
// ct and its mirror in turn and carries the same work.
inline unsigned nt_xcd_row_grid(int64_t rtiles, int nct) { return (unsigned)(8 * nct * ((rtiles + 7) / 8)); }
struct NTXcdRowMap {
  __device__ __forceinline__ int batch_index(unsigned, unsigned by) const { return (int)by; }
  __device__ __forceinline__ void map_block(unsigned bid, int nct, bool loopc, int64_t& rtile, int& ct) const {
    if (loopc) { rtile = bid; ct = 0; return; }
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    rtile = (int64_t)(idx / (unsigned)nct) * 8 + xcd;
    ct = (int)(idx % (unsigned)nct);
#ifndef GDRF_NT_PLAIN_ORDER
    if ((nct & 3) == 0 && ((idx >> 2) & 1u)) ct = (ct & ~3) + 3 - (ct & 3);      // odd group of four: reversed (idx / 4 counts groups across row tiles)
#endif
  }
};

// 1-D grid over (row tile, batch) for workgroups that walk all their column tiles themselves: the nbatch workgroups of
// a row tile (one per topic) run back to back on one XCD and share that row tile's A slab through its L2.
// Grid: 8 * nbatch * ceil(rtiles / 8).
struct NTXcdRowBatchMap {
  int nbatch;
  __device__ __forceinline__ int batch_index(unsigned bid, unsigned) const { return (int)((bid >> 3) % (unsigned)nbatch); }
  __device__ __forceinline__ void map_block(unsigned bid, int, bool, int64_t& rtile, int& ct) const {
    rtile = (int64_t)((bid >> 3) / (unsigned)nbatch) * 8 + (bid & 7u);
    ct = 0;
  }
};

// XCD-aware map for TRIANGULAR reductions, where column tile ct costs (ct+1) units: column tiles are paired
// (ct, nct-1-ct) so that every XCD alternates a light and a heavy tile (equal work per XCD and, with the order flipped every
// four workgroups, per shader engine) while still keeping only two tiles' B panels in its L2.  The XCDs that share a pair deal the row tiles round-robin.
// Grid: nt_xcd_pair_grid(rtiles, nct).  Falls back to the default order when nct is odd or does not divide 8.
__host__ __device__ inline bool nt_xcd_pair_ok(int nct) { return nct >= 2 && nct <= 8 && (nct % 2) == 0 && (8 % nct) == 0; }
inline unsigned nt_xcd_pair_grid(int64_t rtiles, int nct) {
  if (!nt_xcd_pair_ok(nct)) return (unsigned)(rtiles * nct);
  const int nper = 8 / (nct / 2);
  return (unsigned)(16 * ((rtiles + nper - 1) / nper));
}
struct NTXcdPairMap {
  __device__ __forceinline__ int batch_index(unsigned, unsigned by) const { return (int)by; }
  __device__ __forceinline__ void map_block(unsigned bid, int nct, bool loopc, int64_t& rtile, int& ct) const {
    if (loopc) { rtile = bid; ct = 0; return; }
    if (!nt_xcd_pair_ok(nct)) { rtile = bid / nct; ct = (int)(bid % nct); return; }
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    const int c = (int)(xcd % (unsigned)nct), half = nct / 2, nper = 8 / half;
    const int pid = c < half ? c : nct - 1 - c;
    const int r = (int)(xcd / (unsigned)nct) * 2 + (c >= half ? 1 : 0);
    rtile = (int64_t)(idx >> 1) * nper + r;
    ct = ((idx ^ (idx >> 2)) & 1u) ? nct - 1 - pid : pid;   // light, heavy, light, heavy | heavy, light, ...: the shader engine is (idx mod 4), see NTXcdRowMap
  }
};

// row-sum-of-squares epilogue shared by the problems that need it: per-lane partials rs[a][r] over the lane's
// columns -> 16-lane groups -> the two waves that share the rows -> rsum[128] in LDS
template <typename T>
__device__ __forceinline__ void nt_rowsum_finish(T (&rs)[4][4], T* rsum, int wr, int wc, int lane) {
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) rs[a][r] = group16_sum(rs[a][r]);
  if (wc == 0 && (lane & 15) == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) rsum[nt_acc_row<T>(wr, a, lane, r)] = rs[a][r];
  }
  __syncthreads();
  if (wc == 1 && (lane & 15) == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) rsum[nt_acc_row<T>(wr, a, lane, r)] += rs[a][r];
  }
  __syncthreads();
}

}  // namespace gdrf
