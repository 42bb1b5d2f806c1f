// Argument structs shared by the two implementations of the SepConvGRU gate convolutions (sepconv5.hip: direct implicit
// GEMM; sepconv5_wino.hip: 1-D Winograd F(2,5)) -- passed by value as kernel arguments.
#pragma once
#include "common.hpp"

namespace pcfa_sc5 {

// The operand may be the channel concatenation of two tensors ([h | motion features]), read in place.
struct Operand {
  const float* a;
  const float* b;
  int Ca, Cin;
};

// Output channels [0, Ca) go to `a`, the rest to `b` (b may be null when Ca == Cout); acc_*: add to what is there
// (the data gradient of a convolution whose input feeds several consumers accumulates in place instead of being
// summed by separate elementwise kernels).
struct OutSplit {
  float* a;
  float* b;
  int Ca;
  int acc_a, acc_b;
  const float* mask_b;   // optional, shape of `b`: channels < mask_cb of the b part are zeroed where mask_b <= 0
  int mask_cb;           // (the deferred ReLU backward of whoever produced the b operand of the forward)
};

// Fused SepConvGRU epilogues (models/raft/update.py:45-60): the arithmetic of pcfa_gru_gates_fwd / _update_fwd and of
// their backward counterparts applied to the accumulators, so the pre-activations (forward) and the intermediate
// gradients (backward) never reach memory and the elementwise launches between the convolutions disappear.
// All tensors [B][.][H][W]; C % 32 == 0 (a wave's 32 channels are all z or all r, all a-part or all b-part).
//   mode 1, Cout = 2C: m <  C: z[m] = sigmoid(acc + add[m]);  m >= C: r = sigmoid(acc + add[m]), rh = r * h
//           in: p0 = add [B][2C], p1 = h;  out: o0 = z, o1 = r, o2 = r * h
//   mode 2, Cout =  C: q = tanh(acc + add[m]),  hnew = (1 - z) * h + z * q
//           in: p0 = add [B][C], p1 = h, p2 = z;  out: o0 = q, o1 = hnew
//   mode 3 (data gradient of the q convolution, a-part = d(r h)): pcfa_gru_gates_bwd_acc on acc = drh:
//           dzc = dz (1 - z) z,  drc = (drh h)(1 - r) r,  dh = dh_in + drh r
//           in: p0 = z, p1 = r, p2 = h, p3 = dz, p4 = dh_in;  out: o0 = dzr[:, :C], o1 = dzr[:, C:] ([B][2C]), o2 = dh
//   mode 4 (data gradient of the z|r convolution, a-part = dh): g = dh_acc + acc is the gradient of the PREVIOUS
//           half-step's output, whose pcfa_gru_update_bwd follows at once:  dz = g q - g h,  dqc = (g z)(1 - q q),
//           dh = g (1 - z)      in: p0 = dh_acc, p1 = z, p2 = q, p3 = h (previous half);  out: o0 = dz, o1 = dqc, o2 = dh
// Modes 3 / 4 leave the b-part (the motion-feature gradient) to OutSplit; their a-part output pointer is unused.
struct GruEpi {
  int mode, C;
  const float* p0;
  const float* p1;
  const float* p2;
  const float* p3;
  const float* p4;
  float* o0;
  float* o1;
  float* o2;
};

// Floats of the Winograd-domain weights appended to a direct packing of a [Cout][Cin][5] weight (0 when the shape is
// not eligible), and the launcher: returns PCFA_OK after launching, PCFA_SC5_NOT_ELIGIBLE when the direct kernel must run.
constexpr int PCFA_SC5_NOT_ELIGIBLE = -12345;
int sc5_wino_enabled(int set);
bool sc5_wino_shape_ok(int B, int Cin, int Ca, int Cout, int H, int W, int vertical);
long long sc5_wino_packed_floats(int Cout, int Cin);
int sc5_wino_pack(const float* w, float* packed, int Cout, int Cin, int transpose, hipStream_t stream);
int sc5_wino_launch(const Operand& in, const float* w_wino, const OutSplit& out, int B, int Cout, int H, int W,
                    int vertical, hipStream_t stream, const GruEpi& epi);

}  // namespace pcfa_sc5
