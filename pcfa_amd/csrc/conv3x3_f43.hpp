// Internal interface of conv3x3_f43.hip (Winograd F(4x4, 3x3)) used by conv3x3.hip's entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

long long pcfa_f43_packed_floats(int K, int N);
int pcfa_f43_pack(const float* w, float* packed, int Cout, int Cin, int backward, hipStream_t s);
bool pcfa_f43_supported(int B, int K, int N, int H, int W);
int pcfa_f43_ksplit(int B, int K, int N, int H, int W);
size_t pcfa_f43_workspace_bytes(int B, int K, int N, int H, int W);
int pcfa_f43_run(const float* x, const float* packed, const float* bias, const float* mask, const float* addend,
                 float* out, int B, int K, int N, int H, int W, int act, float slope, int mask_n, void* workspace,
                 size_t workspace_bytes, hipStream_t s);
int pcfa_f43_finish(const float* part, const float* bias, const float* mask, const float* addend, float* out, int ksplit,
                    int B, int N, int H, int W, int act, float slope, int mask_n, hipStream_t s, int family = 43);
