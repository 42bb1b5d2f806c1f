// RAFT/GMA correlation-pyramid lookup, forward and backward, for gfx950.
//
// Replaces CorrBlock.__call__ (reference models/raft/corr.py:29-50) and the
// grid_sample inside bilinear_sampler (models/raft/utils/utils.py:57-71).
//
// Data: the pyramid is one slab per query, each level stored as 4x4-texel tiles
// (common.hpp).  All (2r+1)^2 taps of a (query, level) share one fractional
// offset, so a lookup is "fetch a (2r+2)^2 window, blend it 4 ways".
//
// Work decomposition (wave64): workgroup = QB (64 or 32) consecutive queries x one
// level, 2r+1 (or r+1) waves, both directions:
//   * a window is fetched as PIECES: one piece = the 16 B (4 texels of one tile
//     row) that one window row crosses in one tile column, (2r+2) x TXN pieces
//     per window (TXN = tile columns a window can cross).  A wave-instruction
//     moves 64 consecutive pieces of the wave's windows -- whole 64-B sectors,
//     about 11 per window, and nothing but the window's own rows;
//   * the window image lives in LDS with the window's sub-tile offset REMOVED
//     (the shift happens where the image is written / read piecewise, four
//     dwords per lane at a per-lane address), so the per-query side works on
//     aligned rows with 16-B LDS accesses and no per-lane shifting;
//   * forward: pieces -> LDS -> barrier -> thread (query, b) blends the 2r+1
//     taps of window row b; stores run along the query index (256 B per
//     wave-instruction).
//   * backward: the exact transpose with the same ownership: thread (query, row)
//     builds one row of the window's gradient image in LDS, then each piece is
//     read back and added to dpyr (read-modify-write of 16 B per lane).  Every
//     window texel is owned by one lane of one workgroup: no atomics, bitwise
//     reproducible.
#include "common.hpp"

namespace {

constexpr int RS = 20;  // LDS row stride (floats): 4 pad + 16; window texel (r, c) sits at r*RS + 4 + c

#ifndef PCFA_LOOKUP_QB
#define PCFA_LOOKUP_QB 32
#endif

// QB = queries (= windows) per workgroup: 64 (one wave of queries per window row, 2r+1 waves), 32 (two window
// rows per wave, r+1 waves) or 16.  Measured on MI355X at 55x128, r = 4 (rocprofv3, L2-warm / flushed):
// forward 7.4 / 11.9 us at 64, 6.5 / 11.7 us at 32, 7.1-7.5 / 12.4 us at 16; backward 10.6 us at 32 -- more,
// smaller workgroups (880 x 5 waves, 3-4 per CU) overlap each other's load, LDS and store phases better.
template <int R, int QB>
struct Geo {
  static_assert(QB == 64 || QB == 32 || QB == 16, "lane % QB = query needs QB to divide the wave");
  static constexpr int N1 = 2 * R + 1;                      // taps per axis
  static constexpr int WIN = 2 * R + 2;                     // window texels per axis
  static constexpr int RPW = 64 / QB;                       // window rows (phase B) per wave
  static constexpr int NW = (N1 + RPW - 1) / RPW;           // waves per workgroup
  static constexpr int TXN = ((WIN + 2) >> 2) + 1;          // tile columns a window can cross
  static constexpr int PIECES = WIN * TXN;                  // 16-B pieces per window
  static constexpr int NP = (QB * PIECES + NW * 64 - 1) / (NW * 64);  // piece wave-instructions per wave
  static constexpr bool FULL = QB * PIECES == NW * NP * 64; // every piece slot of every wave is a real piece
  static constexpr int WS = WIN * RS + 4;                   // floats per window image; WS/4 odd -> the b128 accesses
                                                            // of 16 consecutive windows hit 16 distinct bank groups
  static constexpr int NRD = (WIN + 3) / 4;                 // b128 accesses per window row
  static constexpr int LDS_FLOATS = QB * WS + 4;            // + a 16-B dump slot for lanes without a piece
};

struct Origin {
  int x0, y0;    // window origin (texels, level coordinates)
  float fx, fy;  // shared bilinear fractions
};

// reference: coords / 2**i  (exact power-of-two scaling), then floor / fraction
__device__ __forceinline__ Origin make_origin(float cx, float cy, int level, int R) {
  const float inv = 1.0f / (float)(1 << level);
  const float xl = cx * inv, yl = cy * inv;
  const float flx = floorf(xl), fly = floorf(yl);
  Origin o;
  o.fx = xl - flx;
  o.fy = yl - fly;
  o.x0 = (int)fminf(fmaxf(flx, -1.0e8f), 1.0e8f) - R;
  o.y0 = (int)fminf(fmaxf(fly, -1.0e8f), 1.0e8f) - R;
  return o;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16-B load at (64-bit scalar base) + (32-bit per-lane byte offset) for the lanes of `mask` only; the other
// lanes keep the previous register contents.  Inline asm so that the compiler neither waits between the
// back-to-back loads of a wave nor turns the mask into a branch; the caller drains them with wait_all_loads()
// before the first use (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void load_piece_masked(f32x4& v, const float* sbase, unsigned voff_bytes,
                                                  unsigned long long mask) {
  unsigned long long save;
  asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %4\n\tglobal_load_dwordx4 %0, %2, %3\n\ts_mov_b64 exec, %1"
               : "+v"(v), "=&s"(save) : "v"(voff_bytes), "s"(sbase), "s"(mask) : "memory");
}

// Pin a wave-uniform pointer into SGPRs (the "s" asm operand above needs it there).
template <typename T>
__device__ __forceinline__ T* scalar_ptr(T* p) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

template <int N>
__device__ __forceinline__ void wait_all_loads(f32x4 (&v)[N]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int k = 0; k < N; ++k) asm volatile("" : "+v"(v[k]));  // uses of v[k] stay below the wait
  __builtin_amdgcn_sched_barrier(0);
}

// Diagnostic stamps (tools/dev/lookup_stamps.py, never in the product kernels): slot k of wave w of workgroup g
// lands in stamps[(g*NW + w)*STAMP_SLOTS + k].  STAMP == false compiles every stamp away.
constexpr int STAMP_SLOTS = 12;
template <bool STAMP>
__device__ __forceinline__ void stamp(unsigned long long* st, int slot, bool drain_vm) {
  if constexpr (STAMP) {
    __builtin_amdgcn_sched_barrier(0);
    if (drain_vm) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    if (threadIdx.x == 0) st[slot] = t;
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <bool STAMP>
__device__ __forceinline__ unsigned long long* stamp_begin(unsigned long long* stamps, int waves, int wv) {
  unsigned long long* st = nullptr;
  if constexpr (STAMP) {
    const int g = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    st = stamps + ((size_t)g * waves + wv) * STAMP_SLOTS;
    unsigned long long rt;
    unsigned hwid, xcc;
    asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\t"
                 "s_waitcnt lgkmcnt(0)" : "=s"(rt), "=s"(hwid), "=s"(xcc)::"memory");
    if (threadIdx.x == 0) {
      st[0] = rt;
      st[10] = ((unsigned long long)xcc << 32) | hwid;
    }
  }
  return st;
}
template <bool STAMP>
__device__ __forceinline__ void stamp_end(unsigned long long* st) {
  if constexpr (STAMP) {
    unsigned long long rt;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
    if (threadIdx.x == 0) st[11] = rt;
  }
}

// Everything a workgroup derives from its (query block, level): wave-uniform layout fields in SGPRs, the
// lane's own query (lane = query index inside the block) and, for the piece loop, each window's parameters
// kept in the registers of the lane that owns the query (fetched cross-lane with ds_bpermute).
template <int R, int QB>
struct Block {
  using G = Geo<R, QB>;
  int level, b_img, q0, hl, wl, tw, lane, wv;
  int j;         // this lane's query inside the block (lane % QB)
  bool live;     // this lane's query exists
  Origin o;      // of this lane's query
  int myB;       // bytes from the workgroup's first slab to this query's level
  int myXY;      // (y0 + 16) | (x0 + 16) << 16, clamped; a window that misses the level collapses onto an
                 // out-of-range origin so that every one of its pieces is invalid

  __device__ __forceinline__ void init(const float* __restrict__ coords, int Q, const PyrLayout& P) {
    level = blockIdx.y;
    b_img = blockIdx.z;
    q0 = blockIdx.x * QB;
    hl = __builtin_amdgcn_readfirstlane(P.h[level]);
    wl = __builtin_amdgcn_readfirstlane(P.w[level]);
    tw = __builtin_amdgcn_readfirstlane(P.tw[level]);
    const int off = __builtin_amdgcn_readfirstlane(P.off[level]);
    const int slab = __builtin_amdgcn_readfirstlane(P.slab);
    lane = threadIdx.x;
    wv = __builtin_amdgcn_readfirstlane(threadIdx.y);  // blockDim.x == 64: one wave per y
    j = lane & (QB - 1);
    live = q0 + j < Q;
    float cx = 0.f, cy = 0.f;
    if (live) {
      cx = coords[((size_t)b_img * 2 + 0) * Q + q0 + j];
      cy = coords[((size_t)b_img * 2 + 1) * Q + q0 + j];
    }
    o = make_origin(cx, cy, level, R);
    const int th4 = ((hl + 3) >> 2) << 2;
    const int x0 = min(max(o.x0, -16), 4 * tw), y0 = live ? min(max(o.y0, -16), th4) : th4;
    myB = (off + j * slab) * 4;
    myXY = (y0 + 16) | ((x0 + 16) << 16);
  }
};

struct Piece {
  unsigned goff;  // bytes from the workgroup's first slab to the piece's 16 B in the pyramid
  unsigned lds;   // byte address in the window image of the piece's first texel (unaligned by the window's ox)
  int mask;       // bit e (< 4): texel e of the piece is a window texel inside the level; bit 4 ("need"): the
                  // piece holds texels of the level at all (else: zeros / nothing to add)
  __device__ __forceinline__ bool need() const { return (mask & 16) != 0; }
};

// Piece slot (wv*NP + i)*64 + lane of the workgroup: window = slot / PIECES (query inside the block), window row
// rr, tile column tx.  The slots of a workgroup are dealt to its waves in order, so a window may be staged by two
// waves.
template <int R, int QB>
__device__ __forceinline__ void fetch_window_params(const Block<R, QB>& blk, int (&wB)[Geo<R, QB>::NP],
                                                    int (&wXY)[Geo<R, QB>::NP]) {
  using G = Geo<R, QB>;
#pragma unroll
  for (int i = 0; i < G::NP; ++i) {  // all cross-lane fetches first: their latencies overlap
    const unsigned src = (((unsigned)blk.wv * G::NP + i) * 64u + (unsigned)blk.lane) / (unsigned)G::PIECES;
    wB[i] = __builtin_amdgcn_ds_bpermute((int)(src * 4u), blk.myB);
    wXY[i] = __builtin_amdgcn_ds_bpermute((int)(src * 4u), blk.myXY);
  }
}

template <int R, int QB>
__device__ __forceinline__ Piece make_piece(const Block<R, QB>& blk, int i, int wB, int wXY) {
  using G = Geo<R, QB>;
  const unsigned p = (((unsigned)blk.wv * G::NP + i) * 64u) + (unsigned)blk.lane;
  const unsigned k = p / (unsigned)G::PIECES, q = p - k * (unsigned)G::PIECES;   // k = query inside the block
  const unsigned rr = q / (unsigned)G::TXN, tx = q - rr * (unsigned)G::TXN;
  const int y0 = (int)((unsigned)wXY & 0xffffu) - 16, x0 = (int)((unsigned)wXY >> 16) - 16;
  const int ox = x0 & 3, y = y0 + (int)rr, gtx = (x0 >> 2) + (int)tx;
  const bool mine = G::FULL || k < (unsigned)QB;
  Piece pc;
  const bool need = mine && (unsigned)y < (unsigned)blk.hl && (unsigned)gtx < (unsigned)blk.tw &&
                    (int)(4 * tx) < ox + G::WIN;
  const int c0 = (int)(4 * tx) - ox, gx0 = 4 * gtx;  // window column / level x of the piece's first texel
  pc.mask = need ? 16 : 0;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if ((unsigned)(c0 + e) < (unsigned)G::WIN && gx0 + e < blk.wl) pc.mask |= 1 << e;
  pc.goff = (unsigned)wB + (((((unsigned)y >> 2) * (unsigned)blk.tw + (unsigned)gtx) << 4) + (((unsigned)y & 3u) << 2)) * 4u;
  const unsigned lds = (k * G::WS + rr * RS + 4u + 4u * tx - (unsigned)ox) * 4u;
  pc.lds = mine ? lds : (unsigned)(QB * G::WS * 4);
  return pc;
}

// STORE: 0 = plain stores, 1 = nontemporal, 2 = write-through (agent-scope relaxed atomic store = sc1)
#ifndef PCFA_LOOKUP_STORE
#define PCFA_LOOKUP_STORE 0
#endif
template <int STORE>
__device__ __forceinline__ void store_out(float* p, float v) {
  if constexpr (STORE == 1) __builtin_nontemporal_store(v, p);
  else if constexpr (STORE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

template <int R, int QB, bool STAMP, int STORE = 0>
__device__ __forceinline__ void corr_lookup_fwd_body(
    const float* __restrict__ pyr, const float* __restrict__ coords, float* __restrict__ out,
    int Q, const PyrLayout& P, unsigned long long* stamps) {
  using G = Geo<R, QB>;
  constexpr int N1 = G::N1, NP = G::NP, NRD = G::NRD;
  __shared__ __attribute__((aligned(16))) float s_win[G::LDS_FLOATS];

  unsigned long long* st = stamp_begin<STAMP>(stamps, G::NW, threadIdx.y);
  stamp<STAMP>(st, 1, false);
  Block<R, QB> blk;
  blk.init(coords, Q, P);
  const float* slab0 = scalar_ptr(pyr + ((size_t)blk.b_img * Q + blk.q0) * P.slab);  // SGPR base of every piece
  stamp<STAMP>(st, 2, true);  // coords landed

  // ---- Phase A: stage the block's windows, 64 pieces per wave-instruction ----
  if (blk.wv * NP * 64 < QB * G::PIECES) {
    f32x4 v[NP];
    unsigned dst[NP];
    int wB[NP], wXY[NP];
    fetch_window_params<R, QB>(blk, wB, wXY);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const Piece pc = make_piece<R, QB>(blk, i, wB[i], wXY[i]);
      dst[i] = pc.lds;
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};  // pieces outside the level are zeros (reference: zero padding)
      load_piece_masked(v[i], slab0, pc.goff, __builtin_amdgcn_ballot_w64(pc.need()));
    }
    stamp<STAMP>(st, 3, false);  // piece loads issued
    wait_all_loads(v);
    stamp<STAMP>(st, 4, false);  // pieces landed
    char* lds_bytes = reinterpret_cast<char*>(s_win);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      float* d = reinterpret_cast<float*>(lds_bytes + dst[i]);
      d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
    }
  }
  stamp<STAMP>(st, 5, false);  // LDS image written
  __syncthreads();
  stamp<STAMP>(st, 6, false);  // barrier passed

  // ---- Phase B: thread (query, b) blends the 2r+1 taps of window row b ---------------------
  const int j = blk.j, b = blk.wv * G::RPW + blk.lane / QB;
  const bool has_row = G::RPW * G::NW == N1 || b < N1;
  const float fx = blk.o.fx, fy = blk.o.fy;
  const int br = has_row ? b : 0;  // lanes beyond the last tap row (QB = 32, odd 2r+1) read row 0 and store nothing
  const float4* row0 = reinterpret_cast<const float4*>(&s_win[j * G::WS + br * RS + 4]);
  const float4* row1 = reinterpret_cast<const float4*>(&s_win[j * G::WS + (br + 1) * RS + 4]);
  float t0[NRD * 4], t1[NRD * 4];
#pragma unroll
  for (int i = 0; i < NRD; ++i) {
    const float4 u0 = row0[i], u1 = row1[i];
    t0[4 * i] = u0.x; t0[4 * i + 1] = u0.y; t0[4 * i + 2] = u0.z; t0[4 * i + 3] = u0.w;
    t1[4 * i] = u1.x; t1[4 * i + 1] = u1.y; t1[4 * i + 2] = u1.z; t1[4 * i + 3] = u1.w;
  }
  const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy);
  const float w10 = (1.f - fx) * fy, w11 = fx * fy;
  const int C = P.L * N1 * N1;
  float* o = out + ((size_t)blk.b_img * C + (size_t)blk.level * N1 * N1 + b) * Q + blk.q0 + j;
  float res[N1];
#pragma unroll
  for (int a = 0; a < N1; ++a) res[a] = t0[a] * w00 + t0[a + 1] * w01 + t1[a] * w10 + t1[a + 1] * w11;
  if constexpr (STAMP) {
#pragma unroll
    for (int a = 0; a < N1; ++a) asm volatile("" : "+v"(res[a]));
  }
  stamp<STAMP>(st, 7, false);  // blended
  if (blk.live && has_row) {
#pragma unroll
    for (int a = 0; a < N1; ++a) store_out<STORE>(o + (size_t)a * N1 * Q, res[a]);
  }
  stamp<STAMP>(st, 8, false);  // stores issued
  stamp<STAMP>(st, 9, true);   // stores acknowledged
  stamp_end<STAMP>(st);
}

template <int R, int QB, bool STAMP>
__device__ __forceinline__ void corr_lookup_bwd_body(
    float* __restrict__ dpyr, const float* __restrict__ coords, const float* __restrict__ grad_out,
    int Q, const PyrLayout& P, unsigned long long* stamps) {
  using G = Geo<R, QB>;
  constexpr int N1 = G::N1, WIN = G::WIN, NP = G::NP, NRD = G::NRD;
  __shared__ __attribute__((aligned(16))) float s_win[G::LDS_FLOATS];

  unsigned long long* st = stamp_begin<STAMP>(stamps, G::NW, threadIdx.y);
  stamp<STAMP>(st, 1, false);
  Block<R, QB> blk;
  blk.init(coords, Q, P);
  float* slab0 = scalar_ptr(dpyr + ((size_t)blk.b_img * Q + blk.q0) * P.slab);
  stamp<STAMP>(st, 2, true);  // coords landed

  // ---- the read half of the read-modify-write goes out first, next to the tap-gradient loads ----
  const bool stager = blk.wv * NP * 64 < QB * G::PIECES;
  f32x4 v[NP];
  Piece pc[NP];
  if (stager) {
    int wB[NP], wXY[NP];
    fetch_window_params<R, QB>(blk, wB, wXY);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      pc[i] = make_piece<R, QB>(blk, i, wB[i], wXY[i]);
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      load_piece_masked(v[i], slab0, pc[i].goff, __builtin_amdgcn_ballot_w64(pc[i].need()));
    }
  }
  stamp<STAMP>(st, 3, false);  // dpyr loads issued

  // ---- Phase A': thread (query, row) builds row `row` of the window's gradient image ----------------
  // d window[r][c] = w00 g[c][r] + w01 g[c-1][r] + w10 g[c][r-1] + w11 g[c-1][r-1]   (g = 0 outside 0..2r)
  const int row = blk.wv * G::RPW + blk.lane / QB;
  if (G::RPW * G::NW == N1 || row < N1) {
    const int j = blk.j;
    const float fx = blk.o.fx, fy = blk.o.fy;
    const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy);
    const float w10 = (1.f - fx) * fy, w11 = fx * fy;
    const int C = P.L * N1 * N1;
    const float* g = grad_out + ((size_t)blk.b_img * C + (size_t)blk.level * N1 * N1) * Q + blk.q0 + j;
    float gc[N1], gp[N1];  // tap gradients of window row `row` and of the row above, along a
#pragma unroll
    for (int a = 0; a < N1; ++a) {
      gc[a] = blk.live ? g[(size_t)(a * N1 + row) * Q] : 0.f;
      gp[a] = (blk.live && row > 0) ? g[(size_t)(a * N1 + row - 1) * Q] : 0.f;
    }
    float4* dst0 = reinterpret_cast<float4*>(&s_win[j * G::WS + row * RS + 4]);
    float d[NRD * 4];
#pragma unroll
    for (int c = 0; c < NRD * 4; ++c) {
      float s = 0.f;
      if (c < N1) s = gc[c] * w00;
      if (c >= 1 && c <= N1) s = fmaf(gc[c - 1], w01, s);
      if (c < N1) s = fmaf(gp[c], w10, s);
      if (c >= 1 && c <= N1) s = fmaf(gp[c - 1], w11, s);
      d[c] = s;
    }
#pragma unroll
    for (int i = 0; i < NRD; ++i) dst0[i] = make_float4(d[4 * i], d[4 * i + 1], d[4 * i + 2], d[4 * i + 3]);
    if (row == N1 - 1) {  // the owner of the last tap row also builds the window's last row (only the row above contributes)
      float4* dst1 = reinterpret_cast<float4*>(&s_win[j * G::WS + (WIN - 1) * RS + 4]);
#pragma unroll
      for (int c = 0; c < NRD * 4; ++c) {
        float s = 0.f;
        if (c < N1) s = gc[c] * w10;
        if (c >= 1 && c <= N1) s = fmaf(gc[c - 1], w11, s);
        d[c] = s;
      }
#pragma unroll
      for (int i = 0; i < NRD; ++i) dst1[i] = make_float4(d[4 * i], d[4 * i + 1], d[4 * i + 2], d[4 * i + 3]);
    }
  }
  stamp<STAMP>(st, 4, false);  // gradient image rows written
  __syncthreads();
  stamp<STAMP>(st, 5, false);  // barrier passed

  // ---- Phase B': every piece adds its 4 texels of the image to dpyr ---------------------------------
  if (stager) {
    wait_all_loads(v);
    stamp<STAMP>(st, 6, false);  // dpyr pieces landed
    const char* lds_bytes = reinterpret_cast<const char*>(s_win);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const float* src = reinterpret_cast<const float*>(lds_bytes + pc[i].lds);
      const float e0 = src[0], e1 = src[1], e2 = src[2], e3 = src[3];
      // image cells outside the window's columns were never written; texels beyond the level width stay zero
      const int m = pc[i].mask;
      f32x4 t = v[i];
      t.x += (m & 1) ? e0 : 0.f;
      t.y += (m & 2) ? e1 : 0.f;
      t.z += (m & 4) ? e2 : 0.f;
      t.w += (m & 8) ? e3 : 0.f;
      if (pc[i].need()) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(slab0) + pc[i].goff) = t;
    }
  }
  stamp<STAMP>(st, 7, false);  // stores issued
  stamp<STAMP>(st, 8, true);   // stores acknowledged
  stamp<STAMP>(st, 9, false);
  stamp_end<STAMP>(st);
}

// __launch_bounds__(1024), not the 64*(2r+1) threads actually launched: with the real bound hipcc derives an
// LDS-limited occupancy (2 workgroups x 9 waves -> 5 waves/SIMD) and INFLATES the kernel descriptor's VGPR
// count to cap the hardware at it (next_free_vgpr 81 for 36 live registers); a 9-wave workgroup then no longer
// fits twice on a CU and the second half of the grid waits for the first (tools/dev/census.hip, measured).
template <int R, int QB = PCFA_LOOKUP_QB>
__global__ __launch_bounds__(1024) void corr_lookup_fwd_kernel(
    const float* __restrict__ pyr, const float* __restrict__ coords, float* __restrict__ out, int Q, PyrLayout P) {
  corr_lookup_fwd_body<R, QB, false, PCFA_LOOKUP_STORE>(pyr, coords, out, Q, P, nullptr);
}

template <int R, int QB = PCFA_LOOKUP_QB>
__global__ __launch_bounds__(1024) void corr_lookup_bwd_kernel(
    float* __restrict__ dpyr, const float* __restrict__ coords, const float* __restrict__ grad_out, int Q,
    PyrLayout P) {
  corr_lookup_bwd_body<R, QB, false>(dpyr, coords, grad_out, Q, P, nullptr);
}

#ifdef PCFA_LOOKUP_DEV
template <int R>
__global__ __launch_bounds__(1024) void corr_lookup_fwd_stamped_kernel(
    const float* __restrict__ pyr, const float* __restrict__ coords, float* __restrict__ out, int Q, PyrLayout P,
    unsigned long long* stamps) {
  corr_lookup_fwd_body<R, PCFA_LOOKUP_QB, true>(pyr, coords, out, Q, P, stamps);
}
template <int STORE>
__global__ __launch_bounds__(1024) void corr_lookup_fwd_store_kernel(
    const float* __restrict__ pyr, const float* __restrict__ coords, float* __restrict__ out, int Q, PyrLayout P) {
  corr_lookup_fwd_body<4, PCFA_LOOKUP_QB, false, STORE>(pyr, coords, out, Q, P, nullptr);
}
template <int R>
__global__ __launch_bounds__(1024) void corr_lookup_bwd_stamped_kernel(
    float* __restrict__ dpyr, const float* __restrict__ coords, const float* __restrict__ grad_out, int Q,
    PyrLayout P, unsigned long long* stamps) {
  corr_lookup_bwd_body<R, PCFA_LOOKUP_QB, true>(dpyr, coords, grad_out, Q, P, stamps);
}
#endif

template <int R>
int launch_fwd(const float* pyr, const float* coords, float* out, int B, int Q,
               const PyrLayout& P, hipStream_t s) {
  using G = Geo<R, PCFA_LOOKUP_QB>;
  dim3 grid(pcfa_cdiv(Q, PCFA_LOOKUP_QB), P.L, B), block(64, G::NW, 1);
  pcfa_launch(corr_lookup_fwd_kernel<R>, grid, block, 0, s, pyr, coords, out, Q, P);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
template <int R>
int launch_bwd(float* dpyr, const float* coords, const float* go, int B, int Q,
               const PyrLayout& P, hipStream_t s) {
  using G = Geo<R, PCFA_LOOKUP_QB>;
  dim3 grid(pcfa_cdiv(Q, PCFA_LOOKUP_QB), P.L, B), block(64, G::NW, 1);
  pcfa_launch(corr_lookup_bwd_kernel<R>, grid, block, 0, s, dpyr, coords, go, Q, P);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

bool check_levels(const PyrLayout& P) {
  for (int l = 0; l < P.L; ++l)
    if (P.h[l] < 1 || P.w[l] < 1 || P.h[l] > 32000 || P.w[l] > 32000) return false;  // packed 16-bit origins
  return true;
}

}  // namespace

extern "C" int pcfa_corr_lookup_fwd(const float* pyr, const float* coords, float* out, int B,
                                    int H, int W, int num_levels, int radius, void* stream) {
  PyrLayout P;
  if (!pyr || !coords || !out || B < 1 || !pcfa_make_layout(P, H, W, num_levels) ||
      !check_levels(P))
    return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int Q = H * W;
  switch (radius) {
    case 1: return launch_fwd<1>(pyr, coords, out, B, Q, P, s);
    case 2: return launch_fwd<2>(pyr, coords, out, B, Q, P, s);
    case 3: return launch_fwd<3>(pyr, coords, out, B, Q, P, s);
    case 4: return launch_fwd<4>(pyr, coords, out, B, Q, P, s);
    default: return PCFA_ERR_UNSUPPORTED;
  }
}

extern "C" int pcfa_corr_lookup_bwd(float* dpyr, const float* coords, const float* grad_out,
                                    int B, int H, int W, int num_levels, int radius,
                                    void* stream) {
  PyrLayout P;
  if (!dpyr || !coords || !grad_out || B < 1 || !pcfa_make_layout(P, H, W, num_levels) ||
      !check_levels(P))
    return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int Q = H * W;
  switch (radius) {
    case 1: return launch_bwd<1>(dpyr, coords, grad_out, B, Q, P, s);
    case 2: return launch_bwd<2>(dpyr, coords, grad_out, B, Q, P, s);
    case 3: return launch_bwd<3>(dpyr, coords, grad_out, B, Q, P, s);
    case 4: return launch_bwd<4>(dpyr, coords, grad_out, B, Q, P, s);
    default: return PCFA_ERR_UNSUPPORTED;
  }
}
