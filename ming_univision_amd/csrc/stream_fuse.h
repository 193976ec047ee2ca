// stream_fuse.h — internal interface of the fused form of the K-slice streaming launch (stream_mfma.hip) used by the RF ResBlock
// chain at <= 4 rows (engine.hip).  Not part of the C ABI.
#pragma once
#include "common.h"

enum { FUSE_NONE = 0, FUSE_SWIGLU = 2 };

// FUSE_SWIGLU: the launch builds its activation image itself, x[m, k] = silu(pb[k] + sum_z pP[z][m][k]) * (pb[K + k] + sum_z pP[z][m][K + k]),
// from the slabs pP [pnz][M][2 * K] of the previous launch and its bias pb [2 * K] (may be NULL).
struct StreamFuse {
  const float* pP; int pnz; const bf16_t* pb;
};

constexpr int FUSE_MAX_ROWS = 4;

// Can this shape run fused (row count, slice length vs threads, slab count of the previous launch)?
bool stream_fused_ok(int wfmt, int M, int Ntot, int K, int prev_nz);
// Launch: W = bf16 [Ntot][K], or e4m3 bytes + wscale (wfmt != 0).  P [nz][M][Ntot], nz = the plain launch's slice count.  Returns nz (< 0: error).
int stream_fused(int wfmt, const void* W, const float* wscale, float* P, int M, int Ntot, int K, const StreamFuse& f, void* stream);
