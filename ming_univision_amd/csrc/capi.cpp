// capi.cpp — host-side plumbing of the C ABI: version, error string, device query.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void mn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mn_version(void) { return MN_VERSION; }
extern "C" const char* mn_last_error(void) { return g_err; }

extern "C" int mn_num_cus(void) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    else
      cus = 256;  // MI355X
  }
  return cus;
}

// ---- tensor-parallel communicator memory (host-side setup, mingnative.h §6) ----------------------------------------------
// Fine-grained allocations are coherent at system scope: peer GPUs' posted writes and this GPU's reads need no cache maintenance
// beyond the release / acquire the kernels already do.
extern "C" int mn_tp_alloc(size_t bytes, void** dptr) {
  MN_CHECK_ARG(dptr && bytes > 0, "mn_tp_alloc: bad args");
  hipError_t e = hipExtMallocWithFlags(dptr, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { mn_set_error("mn_tp_alloc(%zu): %s", bytes, hipGetErrorString(e)); return MN_ELAUNCH; }
  e = hipMemset(*dptr, 0, bytes);
  if (e != hipSuccess) { mn_set_error("mn_tp_alloc memset: %s", hipGetErrorString(e)); return MN_ELAUNCH; }
  return MN_OK;
}
extern "C" int mn_tp_free(void* dptr) { return hipFree(dptr) == hipSuccess ? MN_OK : MN_ELAUNCH; }
extern "C" int mn_tp_ipc_handle(void* dptr, void* handle_out_64) {
  MN_CHECK_ARG(dptr && handle_out_64, "mn_tp_ipc_handle: null pointer");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, dptr);
  if (e != hipSuccess) { mn_set_error("hipIpcGetMemHandle: %s", hipGetErrorString(e)); return MN_ELAUNCH; }
  memcpy(handle_out_64, &h, 64);
  return MN_OK;
}
extern "C" int mn_tp_ipc_open(const void* handle_64, void** dptr) {
  MN_CHECK_ARG(handle_64 && dptr, "mn_tp_ipc_open: null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle_64, 64);
  const hipError_t e = hipIpcOpenMemHandle(dptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) { mn_set_error("hipIpcOpenMemHandle: %s", hipGetErrorString(e)); return MN_ELAUNCH; }
  return MN_OK;
}
extern "C" int mn_tp_ipc_close(void* dptr) { return hipIpcCloseMemHandle(dptr) == hipSuccess ? MN_OK : MN_ELAUNCH; }

// ---- ABI guards: struct sizes and field offsets as THIS build sees them (tests/test_host_logic.py compares them with _lib.py) --------
#include <stddef.h>
extern "C" size_t mn_sizeof_skinny_args(void) { return sizeof(mn_skinny_args); }
extern "C" size_t mn_sizeof_rf_head(void) { return sizeof(mn_rf_head); }
extern "C" size_t mn_sizeof_llm(void) { return sizeof(mn_llm); }
extern "C" size_t mn_sizeof_semdec(void) { return sizeof(mn_semdec); }
extern "C" size_t mn_sizeof_tp_comm(void) { return sizeof(mn_tp_comm); }
extern "C" size_t mn_sizeof_llm_tp(void) { return sizeof(mn_llm_tp); }

#define MN_OFF(T, f) offsetof(T, f),
extern "C" int mn_struct_layout(int which, size_t* offsets, int cap) {
  MN_CHECK_ARG(offsets || cap <= 0, "mn_struct_layout: null offsets");
  static const size_t skinny[] = {
#define F(f) MN_OFF(mn_skinny_args, f)
      F(x) F(ldx) F(w) F(ldw) F(bias) F(out) F(ldo) F(M) F(N) F(K) F(prologue) F(epilogue) F(pro_a) F(ld_pro_a) F(pro_b) F(ld_pro_b)
      F(ln_g) F(ln_b) F(eps) F(res) F(ldres) F(gate) F(ldgate) F(batch) F(w_index) F(w_batch_stride) F(x_batch_stride) F(x_batch_div)
      F(out_batch_stride) F(res_batch_stride) F(nseg) F(seg_index) F(seg_scale) F(seg_w_stride) F(ws) F(ws_bytes) F(wfmt) F(wscale)
      F(wscale_batch_stride) F(wscale_seg_stride)
#undef F
  };
  static const size_t rf[] = {
#define F(f) MN_OFF(mn_rf_head, f)
      F(w) F(depth) F(hidden) F(z_dim) F(target) F(steps) F(llm_hidden) F(vis_w) F(vis_b) F(vis_ln_g) F(vis_ln_b) F(cond_w) F(cond_b)
      F(in_w) F(in_b) F(temb) F(ada_w) F(ada_b) F(ln_g) F(ln_b) F(w12) F(b12) F(w3) F(b3) F(fin_w) F(fin_b) F(wfmt) F(w12_scale)
      F(w3_scale) F(ada_q) F(ada_scale) F(arith)
#undef F
  };
  static const size_t llm[] = {
#define F(f) MN_OFF(mn_llm, f)
      F(hidden) F(n_layers) F(n_q) F(n_kv) F(head_dim) F(n_experts) F(top_k) F(n_shared_slots) F(moe_inter) F(norm_topk_prob) F(rms_eps)
      F(ln1) F(wqkv) F(wdense) F(ln2) F(gate) F(image_gate) F(w_gate_up) F(w_down) F(final_norm) F(cos_tab) F(sin_tab) F(n_pos)
      F(mrope_sec_t) F(mrope_sec_h) F(wfmt) F(w_gate_up_scale) F(w_down_scale) F(arith)
#undef F
  };
  static const size_t semdec[] = {
#define F(f) MN_OFF(mn_semdec, f)
      F(dim) F(depth) F(n_heads) F(hidden) F(in_dim) F(proj_dim) F(proj_depth) F(mean) F(scale) F(in_w) F(in_b) F(ln1_g) F(ln1_b) F(wqkv)
      F(bqkv) F(wproj) F(bproj) F(ln2_g) F(ln2_b) F(w12) F(b12) F(w3) F(b3) F(norm_g) F(norm_b) F(proj_w) F(proj_b) F(hidden_pad) F(w12p)
      F(b12p) F(w3p)
#undef F
  };
  static const size_t comm[] = {
#define F(f) MN_OFF(mn_tp_comm, f)
      F(rank) F(world) F(inbox) F(flags) F(cap) F(rows_cap) F(epoch) F(err) F(wait_ms) F(two_shot_rows)
#undef F
  };
  static const size_t llm_tp[] = {
#define F(f) MN_OFF(mn_llm_tp, f)
      F(expert0) F(n_local_experts) F(shared_inter) F(ws_gate_up) F(ws_down) F(ws_gate_up_scale) F(ws_down_scale)
#undef F
  };
  const size_t* tab[] = {skinny, rf, llm, semdec, comm, llm_tp};
  const int cnt[] = {(int)(sizeof(skinny) / sizeof(size_t)), (int)(sizeof(rf) / sizeof(size_t)), (int)(sizeof(llm) / sizeof(size_t)),
                     (int)(sizeof(semdec) / sizeof(size_t)), (int)(sizeof(comm) / sizeof(size_t)), (int)(sizeof(llm_tp) / sizeof(size_t))};
  MN_CHECK_ARG(which >= 0 && which < 6, "mn_struct_layout: unknown struct id");
  for (int i = 0; i < cnt[which] && i < cap; ++i) offsets[i] = tab[which][i];
  return cnt[which];
}
#undef MN_OFF
