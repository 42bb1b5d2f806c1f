// RAFT/GMA correlation-pyramid lookup, forward and backward, for gfx950.
//
// Replaces CorrBlock.__call__ (reference models/raft/corr.py:29-50) and the
// grid_sample inside bilinear_sampler (models/raft/utils/utils.py:57-71).
//
// Data: the pyramid is one slab per query, each level stored as 4x4-texel tiles
// (common.hpp).  All (2r+1)^2 taps of a (query, level) share one fractional
// offset, so a lookup is "fetch a (2r+2)^2 window, blend it 4 ways".
//
// Work decomposition (wave64): workgroup = 64 consecutive queries x one level,
// 2r+1 waves.
//   A. one WAVE fetches one window per instruction: lane = (tile row, tile col,
//      row in tile) of the 4x4 tile block that covers the window, a single
//      16-B load per lane -> whole 64-B sectors, ~11 per window.  A wave issues
//      the loads of all its windows back to back (8 in flight per lane) before
//      touching LDS.  Window origins are wave-uniform (readlane of the coords),
//      no barrier is needed to share them.
//   B. after one barrier, thread (query, b) reads two window rows from LDS
//      (4 x ds_read_b128 each, conflict-free strides), shifts them by the
//      window's sub-tile offset and blends the 2r+1 taps of row b; stores run
//      along the query index (256-B per wave-instruction).
// The backward is the exact transpose with the same ownership: every window
// texel is owned by one lane of one workgroup, so dpyr += ... needs no atomics
// and is bitwise reproducible.
#include "common.hpp"

namespace {

constexpr int QB = 64;      // queries (= windows) per workgroup
constexpr int WROWS = 13;   // window rows kept in LDS: sub-tile offset (<=3) + 2r+2 (<=10)
constexpr int RS = 20;      // LDS row stride (floats): b128 writes of a tile row block stay conflict-light
constexpr int WS = WROWS * RS;  // 260 floats per window: 260 mod 64 == 4 -> b128 reads of 16 windows conflict-free

template <int R>
struct Geo {
  static constexpr int N1 = 2 * R + 1;
  static constexpr int WIN = 2 * R + 2;
  static constexpr int NWIN = (QB + N1 - 1) / N1;  // windows per wave
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

__device__ __forceinline__ float lane_bcast(float v, int srclane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), srclane));
}

// Element I (compile-time) of a 16-float row held as four float4 -- keeps the row in SSA values so
// the shift below stays a chain of v_cndmask (an indexable array here ends up in scratch memory).
template <int I>
__device__ __forceinline__ float elem16(const float4& a, const float4& b, const float4& c, const float4& d) {
  constexpr int quad = I >> 2, comp = I & 3;
  const float4& v = quad == 0 ? a : quad == 1 ? b : quad == 2 ? c : d;
  return comp == 0 ? v.x : comp == 1 ? v.y : comp == 2 ? v.z : v.w;
}

template <int WIN, int C>
struct ShiftRow {
  __device__ __forceinline__ static void run(const float4& a, const float4& b, const float4& c, const float4& d,
                                             bool by2, bool by1, float* t) {
    // element ox + C of the row, ox = 2*by2 + by1
    const float e0 = by2 ? elem16<C + 2>(a, b, c, d) : elem16<C>(a, b, c, d);
    const float e1 = by2 ? elem16<C + 3>(a, b, c, d) : elem16<C + 1>(a, b, c, d);
    t[C] = by1 ? e1 : e0;
    ShiftRow<WIN, C + 1>::run(a, b, c, d, by2, by1, t);
  }
};
template <int WIN>
struct ShiftRow<WIN, WIN> {
  __device__ __forceinline__ static void run(const float4&, const float4&, const float4&, const float4&, bool,
                                             bool, float*) {}
};

// t[c] = row[ox + c], c < WIN, ox in 0..3: four aligned ds_read_b128 + selects (no unaligned LDS access)
template <int WIN>
__device__ __forceinline__ void shifted_row(const float* row, int ox, float* t) {
  const float4 a = *reinterpret_cast<const float4*>(row);
  const float4 b = *reinterpret_cast<const float4*>(row + 4);
  const float4 c = *reinterpret_cast<const float4*>(row + 8);
  const float4 d = *reinterpret_cast<const float4*>(row + 12);
  ShiftRow<WIN, 0>::run(a, b, c, d, (ox & 2) != 0, (ox & 1) != 0, t);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16-B load at (64-bit scalar base) + (32-bit per-lane byte offset).  Inline asm so that the compiler
// neither predicates it back into a branch nor waits between the back-to-back loads of a wave; the
// caller drains them with wait_all_loads() before the first use (cdna_hip_programming.md 5.7).
__device__ __forceinline__ f32x4 load_tile_row(const float* sbase, unsigned voff_bytes) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff_bytes), "s"(sbase) : "memory");
  return v;
}

// Same load for the lanes of `mask` only; the other lanes keep the previous register contents.
__device__ __forceinline__ void load_tile_row_masked(f32x4& v, const float* sbase, unsigned voff_bytes,
                                                     unsigned long long mask) {
  unsigned long long save;
  asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %4\n\tglobal_load_dwordx4 %0, %2, %3\n\ts_mov_b64 exec, %1"
               : "+v"(v), "=&s"(save) : "v"(voff_bytes), "s"(sbase), "s"(mask) : "memory");
}

// Pin a wave-uniform pointer into SGPRs (the "s" asm operand above needs it there).
__device__ __forceinline__ const float* scalar_ptr(const float* p) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}

template <int N>
__device__ __forceinline__ void wait_all_loads(f32x4 (&v)[N]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int k = 0; k < N; ++k) asm volatile("" : "+v"(v[k]));  // uses of v[k] stay below the wait
  __builtin_amdgcn_sched_barrier(0);
}

// Per-window scalar bookkeeping shared by forward and backward: which lanes of the 4x4-tile block
// (lane = (ty, tx, r), row ry = 4*ty + r) hold texels of the window, and where the block starts.
struct WindowBlock {
  int tile_off;        // floats from the slab's level start to tile (ty0, tx0); valid lanes only
  int lo_tx, n_tx;     // needed tile columns: tx - lo_tx < n_tx   (unsigned compare)
  int lo_ry, n_ry;     // needed rows:         ry - lo_ry < n_ry
};

template <int WIN>
__device__ __forceinline__ WindowBlock window_block(int x0, int y0, int tw, int hrows, bool valid) {
  const int tx0 = x0 >> 2, ty0 = y0 >> 2;
  WindowBlock w;
  w.tile_off = (ty0 * tw + tx0) * 16;
  const int lo_t = max(0, tx0), hi_t = min(tw, ((x0 + WIN - 1) >> 2) + 1);
  const int lo_y = max(0, y0), hi_y = min(hrows, y0 + WIN);
  const bool hit = valid && hi_t > lo_t && hi_y > lo_y;  // window intersects the level at all
  // (far-away / NaN coordinates: keep every field inside its packed bit range)
  w.lo_tx = hit ? lo_t - tx0 : 0;      // 0..3
  w.n_tx = hit ? hi_t - lo_t : 0;      // 0..4
  w.lo_ry = hit ? lo_y - ty0 * 4 : 0;  // 0..12
  w.n_ry = hit ? hi_y - lo_y : 0;      // 0..WIN
  if (!hit) w.tile_off = 0;
  return w;
}

// Diagnostic stamps (tools/dev/lookup_stamps.*, never in the product kernel): slot k of wave w of workgroup g
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

// Window image in LDS (forward): window j occupies WS2 floats = WIN rows of RS2 floats; row r holds window
// texel (r, c) at column 4 + c -- the sub-tile offset (ox, oy) of the window inside its 16x16 fetch block is
// removed when the fetched tile rows are WRITTEN (a wave-uniform address shift, four dword stores per lane),
// so phase B reads aligned rows with ds_read_b128 and needs no per-lane shifting.  WS2/4 is odd: the b128
// reads of 16 consecutive windows are bank-conflict free.
constexpr int RS2 = 20;
template <int R>
struct Geo2 {
  static constexpr int WIN = 2 * R + 2;
  static constexpr int WS = WIN * RS2 + 4;
  static constexpr int NRD = (WIN + 3) / 4;  // b128 reads per window row
};

template <int R, bool STAMP, bool MASKED_LOADS>
__device__ __forceinline__ void corr_lookup_fwd_body(
    const float* __restrict__ pyr, const float* __restrict__ coords, float* __restrict__ out,
    int Q, int qb, const PyrLayout& P, unsigned long long* stamps) {
  using G = Geo<R>;
  constexpr int N1 = G::N1, WIN = G::WIN, NWIN = G::NWIN;
  constexpr int WS2 = Geo2<R>::WS, NRD = Geo2<R>::NRD;
  __shared__ __attribute__((aligned(16))) float s_win[QB * WS2];
  __shared__ float s_fx[QB], s_fy[QB];

  const int level = blockIdx.y, b_img = blockIdx.z;
  const int q0 = blockIdx.x * qb;  // qb <= QB queries per workgroup
  // wave-uniform layout fields, pinned to SGPRs before any divergent branch
  const int hl = __builtin_amdgcn_readfirstlane(P.h[level]);
  const int tw = __builtin_amdgcn_readfirstlane(P.tw[level]);
  const int off = __builtin_amdgcn_readfirstlane(P.off[level]);
  const int slab = __builtin_amdgcn_readfirstlane(P.slab);
  const unsigned zero4 = (unsigned)__builtin_amdgcn_readfirstlane(P.zero) * 4u;
  const int th4 = ((hl + 3) >> 2) << 2;  // padded height (pad rows hold zeros)
  const int lane = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);  // blockDim.x == 64: one wave per y
  const float* slab0 = scalar_ptr(pyr + ((size_t)b_img * Q + q0) * slab);  // SGPR base of every window load
  unsigned long long* st = nullptr;
  if constexpr (STAMP) {
    const int g = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    st = stamps + ((size_t)g * N1 + wv) * STAMP_SLOTS;
    unsigned long long rt;
    unsigned hwid, xcc;
    asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\t"
                 "s_waitcnt lgkmcnt(0)" : "=s"(rt), "=s"(hwid), "=s"(xcc)::"memory");
    if (lane == 0) { st[0] = rt; st[10] = ((unsigned long long)xcc << 32) | hwid; }
  }
  stamp<STAMP>(st, 1, false);

  // ---- Phase A: window fetch, one window per wave-instruction -----------------------------
  // Lane k (< NWIN) does ALL the bookkeeping of this wave's k-th window (j = wv + k*N1) once, in
  // vector registers; the loop below only broadcasts two packed words per window (v_readlane).
  int myT = 0, myP = 0;
  {
    const int j = wv + lane * N1;
    if (lane < NWIN && j < qb) {
      const bool valid = q0 + j < Q;
      float cx = 0.f, cy = 0.f;
      if (valid) {
        cx = coords[((size_t)b_img * 2 + 0) * Q + q0 + j];
        cy = coords[((size_t)b_img * 2 + 1) * Q + q0 + j];
      }
      const Origin o = make_origin(cx, cy, level, R);
      s_fx[j] = o.fx;
      s_fy[j] = o.fy;
      const WindowBlock w = window_block<WIN>(o.x0, o.y0, tw, th4, valid);
      myT = (off + w.tile_off + (valid ? j : 0) * slab) * 4;  // bytes from this workgroup's first slab
      // bits 0-3 lo_tx, 4-7 n_tx, 8-11 lo_ry, 12-15 n_ry, 16-25 LDS shift (oy*RS2 + ox)*4 bytes
      myP = w.lo_tx | (w.n_tx << 4) | (w.lo_ry << 8) | (w.n_ry << 12) |
            ((((o.y0 & 3) * RS2 + (o.x0 & 3)) * 4) << 16) | ((o.y0 & 3) << 28);
    }
  }
  stamp<STAMP>(st, 2, true);  // coords landed, bookkeeping done
  const int ty = lane >> 4, tx = (lane >> 2) & 3, r = lane & 3;
  const int ry = ty * 4 + r;                                      // row inside the 16x16 texel block
  const int lane_goff4 = (((ty * tw + tx) << 4) + (r << 2)) * 4;  // bytes, relative to tile (ty0, tx0)
  // LDS byte address of this lane's 4 texels for a window with ox = oy = 0 (window k adds k*N1*WS2, minus the shift)
  const unsigned lane_lds = (unsigned)((wv * WS2 + ry * RS2 + 4 + tx * 4) * 4);
  f32x4 v[NWIN];
#pragma unroll
  for (int k = 0; k < NWIN; ++k) {
    const int sT = __builtin_amdgcn_readlane(myT, k), sP = __builtin_amdgcn_readlane(myP, k);
    const bool need = (unsigned)(tx - (sP & 15)) < (unsigned)((sP >> 4) & 15) &&
                      (unsigned)(ry - ((sP >> 8) & 15)) < (unsigned)((sP >> 12) & 15);
    if constexpr (MASKED_LOADS) {
      v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      load_tile_row_masked(v[k], slab0, (unsigned)(sT + lane_goff4), __builtin_amdgcn_ballot_w64(need));
    } else {
      // lanes outside the window read the all-zero tile of slab q0: branch-free, no select afterwards
      v[k] = load_tile_row(slab0, need ? (unsigned)(sT + lane_goff4) : zero4);
    }
  }
  stamp<STAMP>(st, 3, false);  // window loads issued
  wait_all_loads(v);
  stamp<STAMP>(st, 4, false);  // windows landed
  char* lds_bytes = reinterpret_cast<char*>(s_win);
#pragma unroll
  for (int k = 0; k < NWIN; ++k) {
    const int j = wv + k * N1;
    const int sP = __builtin_amdgcn_readlane(myP, k);
    const int oy = (sP >> 28) & 3;
    if (j < qb && (unsigned)(ry - oy) < (unsigned)WIN) {
      float* dst = reinterpret_cast<float*>(lds_bytes + (lane_lds + (unsigned)(k * N1 * WS2 * 4) - (unsigned)((sP >> 16) & 1023)));
      dst[0] = v[k].x; dst[1] = v[k].y; dst[2] = v[k].z; dst[3] = v[k].w;
    }
  }
  stamp<STAMP>(st, 5, false);  // LDS image written
  __syncthreads();
  stamp<STAMP>(st, 6, false);  // barrier passed

  // ---- Phase B: thread (query, b) blends the 2r+1 taps of window row b ---------------------
  const int j = lane, b = wv;
  if constexpr (!STAMP) {
    if (j >= qb || q0 + j >= Q) return;
  }
  const bool live = j < qb && q0 + j < Q;
  const float fx = s_fx[j], fy = s_fy[j];
  const float4* row0 = reinterpret_cast<const float4*>(&s_win[j * WS2 + b * RS2 + 4]);
  const float4* row1 = reinterpret_cast<const float4*>(&s_win[j * WS2 + (b + 1) * RS2 + 4]);
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
  float* o = out + ((size_t)b_img * C + (size_t)level * N1 * N1 + b) * Q + q0 + j;
  float res[N1];
#pragma unroll
  for (int a = 0; a < N1; ++a) res[a] = t0[a] * w00 + t0[a + 1] * w01 + t1[a] * w10 + t1[a + 1] * w11;
  if constexpr (STAMP) {
#pragma unroll
    for (int a = 0; a < N1; ++a) asm volatile("" : "+v"(res[a]));
  }
  stamp<STAMP>(st, 7, false);  // blended
  if (live) {
#pragma unroll
    for (int a = 0; a < N1; ++a) o[(size_t)a * N1 * Q] = res[a];
  }
  stamp<STAMP>(st, 8, false);  // stores issued
  stamp<STAMP>(st, 9, true);   // stores acknowledged
  if constexpr (STAMP) {
    unsigned long long rt;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
    if (lane == 0) st[11] = rt;
  }
}

#ifdef PCFA_LOOKUP_DEV
template <int R, bool STAMP>
__device__ __forceinline__ void corr_lookup_fwd_body_v1(
    const float* __restrict__ pyr, const float* __restrict__ coords, float* __restrict__ out,
    int Q, int qb, const PyrLayout& P, unsigned long long* stamps) {
  using G = Geo<R>;
  constexpr int N1 = G::N1, WIN = G::WIN, NWIN = G::NWIN;
  __shared__ __attribute__((aligned(16))) float s_win[QB * WS];
  __shared__ int s_ox[QB], s_oy[QB];
  __shared__ float s_fx[QB], s_fy[QB];

  const int level = blockIdx.y, b_img = blockIdx.z;
  const int q0 = blockIdx.x * qb;  // qb <= QB queries per workgroup, chosen on the host for an even CU load
  // wave-uniform layout fields, pinned to SGPRs before any divergent branch
  const int hl = __builtin_amdgcn_readfirstlane(P.h[level]);
  const int tw = __builtin_amdgcn_readfirstlane(P.tw[level]);
  const int off = __builtin_amdgcn_readfirstlane(P.off[level]);
  const int slab = __builtin_amdgcn_readfirstlane(P.slab);
  const unsigned zero4 = (unsigned)__builtin_amdgcn_readfirstlane(P.zero) * 4u;
  const int th4 = ((hl + 3) >> 2) << 2;  // padded height (pad rows hold zeros)
  const int lane = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);  // blockDim.x == 64: one wave per y
  const float* slab0 = scalar_ptr(pyr + ((size_t)b_img * Q + q0) * slab);  // SGPR base of every window load
  unsigned long long* st = nullptr;
  if constexpr (STAMP) {
    const int g = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    st = stamps + ((size_t)g * N1 + wv) * STAMP_SLOTS;
    unsigned long long rt;
    unsigned hwid, xcc;
    asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\t"
                 "s_waitcnt lgkmcnt(0)" : "=s"(rt), "=s"(hwid), "=s"(xcc)::"memory");
    if (lane == 0) { st[0] = rt; st[10] = ((unsigned long long)xcc << 32) | hwid; }
  }
  stamp<STAMP>(st, 1, false);

  // ---- Phase A: window fetch, one window per wave-instruction -----------------------------
  // Lane k (< NWIN) does ALL the bookkeeping of this wave's k-th window (j = wv + k*N1) once, in
  // vector registers; the loop below only broadcasts two packed words per window (v_readlane).
  int myT = 0, myP = 0;
  {
    const int j = wv + lane * N1;
    if (lane < NWIN && j < qb) {
      const bool valid = q0 + j < Q;
      float cx = 0.f, cy = 0.f;
      if (valid) {
        cx = coords[((size_t)b_img * 2 + 0) * Q + q0 + j];
        cy = coords[((size_t)b_img * 2 + 1) * Q + q0 + j];
      }
      const Origin o = make_origin(cx, cy, level, R);
      s_ox[j] = o.x0 & 3;  // offset of the window inside its first tile (floor mod)
      s_oy[j] = o.y0 & 3;
      s_fx[j] = o.fx;
      s_fy[j] = o.fy;
      const WindowBlock w = window_block<WIN>(o.x0, o.y0, tw, th4, valid);
      myT = (off + w.tile_off + (valid ? j : 0) * slab) * 4;  // bytes from this workgroup's first slab
      myP = w.lo_tx | (w.n_tx << 4) | (w.lo_ry << 8) | (w.n_ry << 16);
    }
  }
  stamp<STAMP>(st, 2, true);  // coords landed, bookkeeping done
  const int ty = lane >> 4, tx = (lane >> 2) & 3, r = lane & 3;
  const int ry = ty * 4 + r;                                      // row inside the 16x16 texel block
  const int lane_goff4 = (((ty * tw + tx) << 4) + (r << 2)) * 4;  // bytes, relative to tile (ty0, tx0)
  float* lds_row = &s_win[wv * WS + ry * RS + tx * 4];            // window j = wv + k*N1 adds k*N1*WS
  f32x4 v[NWIN];
#pragma unroll
  for (int k = 0; k < NWIN; ++k) {
    const int sT = __builtin_amdgcn_readlane(myT, k), sP = __builtin_amdgcn_readlane(myP, k);
    const bool need = (unsigned)(tx - (sP & 15)) < (unsigned)((sP >> 4) & 15) &&
                      (unsigned)(ry - ((sP >> 8) & 255)) < (unsigned)(sP >> 16);
    // lanes outside the window read the all-zero tile of slab q0: branch-free, no select afterwards
    v[k] = load_tile_row(slab0, need ? (unsigned)(sT + lane_goff4) : zero4);
  }
  stamp<STAMP>(st, 3, false);  // window loads issued
  wait_all_loads(v);
  stamp<STAMP>(st, 4, false);  // windows landed
#pragma unroll
  for (int k = 0; k < NWIN; ++k) {
    const int j = wv + k * N1;
    if (j < qb && ry < WROWS) *reinterpret_cast<f32x4*>(lds_row + k * N1 * WS) = v[k];
  }
  stamp<STAMP>(st, 5, false);  // LDS image written
  __syncthreads();
  stamp<STAMP>(st, 6, false);  // barrier passed

  // ---- Phase B: thread (query, b) blends the 2r+1 taps of window row b ---------------------
  const int j = lane, b = wv;
  if constexpr (!STAMP) {
    if (j >= qb || q0 + j >= Q) return;
  }
  const bool live = j < qb && q0 + j < Q;
  const int ox = s_ox[j], oy = s_oy[j];
  const float fx = s_fx[j], fy = s_fy[j];
  const float* row0 = &s_win[j * WS + (oy + b) * RS];
  float t0[WIN], t1[WIN];
  shifted_row<WIN>(row0, ox, t0);
  shifted_row<WIN>(row0 + RS, ox, t1);
  const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy);
  const float w10 = (1.f - fx) * fy, w11 = fx * fy;
  const int C = P.L * N1 * N1;
  float* o = out + ((size_t)b_img * C + (size_t)level * N1 * N1 + b) * Q + q0 + j;
  float res[N1];
#pragma unroll
  for (int a = 0; a < N1; ++a) res[a] = t0[a] * w00 + t0[a + 1] * w01 + t1[a] * w10 + t1[a + 1] * w11;
  if constexpr (STAMP) {
#pragma unroll
    for (int a = 0; a < N1; ++a) asm volatile("" : "+v"(res[a]));
  }
  stamp<STAMP>(st, 7, false);  // blended
  if (live) {
#pragma unroll
    for (int a = 0; a < N1; ++a) o[(size_t)a * N1 * Q] = res[a];
  }
  stamp<STAMP>(st, 8, false);  // stores issued
  stamp<STAMP>(st, 9, true);   // stores acknowledged
  if constexpr (STAMP) {
    unsigned long long rt;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
    if (lane == 0) st[11] = rt;
  }
}

#endif

// __launch_bounds__(1024), not the 64*(2r+1) threads actually launched: with the real bound hipcc derives an
// LDS-limited occupancy (2 workgroups x 9 waves -> 5 waves/SIMD) and INFLATES the kernel descriptor's VGPR
// count to cap the hardware at it (next_free_vgpr 81 for 36 live registers); a 9-wave workgroup then no longer
// fits twice on a CU and the second half of the grid waits for the first (tools/dev/census.hip, measured).
template <int R>
__global__ __launch_bounds__(1024) void corr_lookup_fwd_kernel(
    const float* __restrict__ pyr, const float* __restrict__ coords, float* __restrict__ out,
    int Q, int qb, PyrLayout P) {
  corr_lookup_fwd_body<R, false, false>(pyr, coords, out, Q, qb, P, nullptr);
}

#ifdef PCFA_LOOKUP_DEV
// A/B variants for tools/dev: VAR 0 = previous kernel (bounded), 1 = previous kernel with the 1024 bound,
// 2 = this kernel, 3 = this kernel with exec-masked window loads instead of zero-tile loads.
template <int VAR, bool STAMP>
__global__ __launch_bounds__(VAR == 0 ? 576 : 1024) void corr_lookup_fwd_var_kernel(
    const float* __restrict__ pyr, const float* __restrict__ coords, float* __restrict__ out,
    int Q, int qb, PyrLayout P, unsigned long long* stamps) {
  if constexpr (VAR <= 1) corr_lookup_fwd_body_v1<4, STAMP>(pyr, coords, out, Q, qb, P, stamps);
  else corr_lookup_fwd_body<4, STAMP, VAR == 3>(pyr, coords, out, Q, qb, P, stamps);
}
#endif

template <int R>
__global__ __launch_bounds__(QB*(2 * R + 1)) void corr_lookup_bwd_kernel(
    float* __restrict__ dpyr, const float* __restrict__ coords,
    const float* __restrict__ grad_out, int Q, PyrLayout P) {
  using G = Geo<R>;
  constexpr int N1 = G::N1, WIN = G::WIN, NWIN = G::NWIN;
  constexpr int GS = N1 * N1 + ((N1 * N1) % 2 == 0 ? 1 : 0);  // odd stride
  __shared__ float s_g[QB * GS];

  const int level = blockIdx.y, b_img = blockIdx.z;
  const int q0 = blockIdx.x * QB;
  const int hl = P.h[level], wl = P.w[level], tw = P.tw[level], off = P.off[level];
  const int lane = threadIdx.x, wv = threadIdx.y;

  // ---- Phase A': tap gradients, lanes along the query index (coalesced) --------------------
  {
    const int q = q0 + lane;
    const int C = P.L * N1 * N1;
    const float* g = grad_out + ((size_t)b_img * C + (size_t)level * N1 * N1 + wv) * Q + q;
    float gv[N1];
#pragma unroll
    for (int a = 0; a < N1; ++a) gv[a] = (q < Q) ? g[(size_t)a * N1 * Q] : 0.f;
    __builtin_amdgcn_sched_barrier(0);  // keep the 2r+1 loads in flight together
#pragma unroll
    for (int a = 0; a < N1; ++a) s_g[lane * GS + wv * N1 + a] = gv[a];
  }
  float mycx = 0.f, mycy = 0.f;
  {
    const int j = wv + lane * N1;
    if (lane < NWIN && j < QB && q0 + j < Q) {
      mycx = coords[((size_t)b_img * 2 + 0) * Q + q0 + j];
      mycy = coords[((size_t)b_img * 2 + 1) * Q + q0 + j];
    }
  }
  __syncthreads();

  // ---- Phase B': one window per wave-instruction; a lane owns 4 texels of a tile row --------
  float* base = dpyr + (size_t)b_img * Q * P.slab + off;
  const int ty = lane >> 4, tx = (lane >> 2) & 3, r = lane & 3;
  const int ry = ty * 4 + r;
  float4 v[NWIN];
  float* ptr[NWIN];
  Origin org[NWIN];
#pragma unroll
  for (int k = 0; k < NWIN; ++k) {
    const int j = wv + k * N1;
    org[k] = make_origin(lane_bcast(mycx, k), lane_bcast(mycy, k), level, R);
    const int tx0 = org[k].x0 >> 2, ty0 = org[k].y0 >> 2;
    const int gy = ty0 * 4 + ry, gtx = tx0 + tx;
    const bool need = (j < QB) && (q0 + j < Q) && gtx >= 0 && gtx < tw && gy >= 0 && gy < hl &&
                      gy >= org[k].y0 && gy < org[k].y0 + WIN && gtx * 4 + 3 >= org[k].x0 &&
                      gtx * 4 < org[k].x0 + WIN;
    ptr[k] = need ? base + (size_t)(q0 + j) * P.slab + (((gy >> 2) * tw + gtx) << 4) + ((gy & 3) << 2)
                  : nullptr;
    // branch-free load (dummy lanes read this workgroup's first sector) so all NWIN loads are in flight
    v[k] = *reinterpret_cast<const float4*>(need ? ptr[k] : base + (size_t)q0 * P.slab);
  }
  __builtin_amdgcn_sched_barrier(0);  // all read-modify-write loads issued before the gather math
#pragma unroll
  for (int k = 0; k < NWIN; ++k) {
    if (ptr[k] == nullptr) continue;
    const int j = wv + k * N1;
    const float fx = org[k].fx, fy = org[k].fy;
    const float* g = s_g + j * GS;
    const int tx0 = org[k].x0 >> 2, ty0 = org[k].y0 >> 2;
    const int rr = ty0 * 4 + ry - org[k].y0;  // window row of this lane's texels, 0..WIN-1
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int gx = (tx0 + tx) * 4 + e;
      const int cc = gx - org[k].x0;  // window column
      float s = 0.f;
      if (cc >= 0 && cc < WIN && gx < wl) {
        // tap (a, b) touches window texels (row, col) in {b, b+1} x {a, a+1}
        if (rr < N1) {
          const float wy = 1.f - fy;
          if (cc < N1) s += g[rr * N1 + cc] * ((1.f - fx) * wy);
          if (cc > 0) s += g[rr * N1 + cc - 1] * (fx * wy);
        }
        if (rr > 0) {
          const float wy = fy;
          if (cc < N1) s += g[(rr - 1) * N1 + cc] * ((1.f - fx) * wy);
          if (cc > 0) s += g[(rr - 1) * N1 + cc - 1] * (fx * wy);
        }
      }
      acc[e] = s;
    }
    float4 t = v[k];
    t.x += acc[0]; t.y += acc[1]; t.z += acc[2]; t.w += acc[3];
    *reinterpret_cast<float4*>(ptr[k]) = t;
  }
}

// Queries per workgroup.  64 (= one wave of queries in phase B) measured best on MI355X at Q = 7040, 4 levels:
// 440 workgroups, 9.9 us; a CU-balanced 55 (512 workgroups, exactly 2 per CU) was SLOWER, 10.8 us -- the kernel is
// bound by per-workgroup latency, not by the most loaded CU.
int balanced_queries_per_group(int /*Q*/, int /*planes*/) { return QB; }

template <int R>
int launch_fwd(const float* pyr, const float* coords, float* out, int B, int Q,
               const PyrLayout& P, hipStream_t s) {
  const int qb = balanced_queries_per_group(Q, P.L * B);
  dim3 grid(pcfa_cdiv(Q, qb), P.L, B), block(QB, 2 * R + 1, 1);
  pcfa_launch(corr_lookup_fwd_kernel<R>, grid, block, 0, s, pyr, coords, out, Q, qb, P);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
template <int R>
int launch_bwd(float* dpyr, const float* coords, const float* go, int B, int Q,
               const PyrLayout& P, hipStream_t s) {
  dim3 grid(pcfa_cdiv(Q, QB), P.L, B), block(QB, 2 * R + 1, 1);
  pcfa_launch(corr_lookup_bwd_kernel<R>, grid, block, 0, s, dpyr, coords, go, Q, P);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

bool check_levels(const PyrLayout& P) {
  for (int l = 0; l < P.L; ++l)
    if (P.h[l] < 1 || P.w[l] < 1) return false;
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
