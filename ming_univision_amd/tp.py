"""Tensor / expert parallelism of the decode path over the GPUs of one xGMI node (BASELINE configs[4]).

No reference counterpart: the reference runs Ming-UniVision-16B-A3B on one device.  One process per GPU; every rank holds a SHARD
of the decoder stack and of the RF head and calls the same composites (`mn_llm_step_tp`, `mn_rf_sample_tp`) with the same
arguments; MingTok (0.7 B parameters) and the small replicated layers run on every rank.

  attention   q heads split over the ranks, KV head h on the ranks whose q heads belong to it (16 q / 4 KV heads at TP = 8:
              2 q heads + 1 KV head per rank, each KV head on 2 ranks); query_key_value row-split, dense column-split
  experts     E / world routed experts per rank, every rank keeps all rows + the global routing (replicate-and-reduce EP);
              the shared expert's intermediate width is split world ways (zero-padded to a multiple of 64)
  RF head     SwiGLU hidden width split world ways (w12 by rows, w3 by columns)
  all-reduce  one-shot push over xGMI into per-rank inboxes + arrival flags, consumed by the next row kernel (csrc/tp.inl);
              2 per decoder layer, 1 per ResBlock per Euler step

Host side here: the partitioning of reference-named / packed weights (pure indexing, also exercised on CPU tensors by the
world-2 gloo test), the communicator setup (fine-grained buffers + IPC handles exchanged through torch.distributed — RCCL on GPUs,
gloo in the CPU test), the per-rank shard objects, and `TpSimGroup`: all ranks' shards in ONE process on ONE GPU, advanced segment
by segment — the parity harness of a path whose hardware (8 GPUs) the build container does not have.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import Llm, LlmTp, RfHead, TpComm, check, current_stream, lib, ptr, ptr_array


# ------------------------------------------------------------------------------------------------------------------------
# partitioning (pure indexing: works on CPU and GPU tensors)
# ------------------------------------------------------------------------------------------------------------------------
def shard_plan(cfg, world, rf_hidden=None):
    """Checks that the 16B-A3B-style configuration splits `world` ways and returns the per-rank sizes."""
    nq, nkv, E = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.num_experts
    S, I = cfg.num_shared_experts or 0, cfg.moe_intermediate_size
    if nq % world:
        raise ValueError(f"{nq} query heads do not split {world} ways")
    if not (nkv % world == 0 or world % nkv == 0):
        raise ValueError(f"{nkv} KV heads neither split nor replicate over {world} ranks")
    if (nq // world) * max(1, world // nkv) > nq // nkv and nkv % world:
        raise ValueError("a rank's query heads would span two KV heads")
    if E % world:
        raise ValueError(f"{E} experts do not split {world} ways")
    if (S * I) % world:
        raise ValueError(f"shared intermediate width {S * I} does not split {world} ways")
    if rf_hidden is not None and (rf_hidden % world or (rf_hidden // world) % 64):
        raise ValueError(f"RF SwiGLU width {rf_hidden} does not split {world} ways into multiples of 64")
    sh = S * I // world
    return dict(n_q=nq // world, n_kv=max(1, nkv // world), n_experts=E // world, shared=sh, shared_pad=(sh + 63) // 64 * 64 if sh else 0,
                rf_hidden=None if rf_hidden is None else rf_hidden // world)


def shard_attention(wqkv, wdense, cfg, rank, world):
    """query_key_value [(nq + 2 nkv) hd, H] -> this rank's rows [q heads | its K | its V]; dense [H, nq hd] -> its columns."""
    nq, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    pl = shard_plan(cfg, world)
    q0 = rank * pl["n_q"]
    kv0 = (q0 * nkv) // nq                         # KV head of this rank's first q head (GQA: q head i uses KV head i // (nq / nkv))
    rows = torch.cat([torch.arange(q0 * hd, (q0 + pl["n_q"]) * hd),
                      torch.arange((nq + kv0) * hd, (nq + kv0 + pl["n_kv"]) * hd),
                      torch.arange((nq + nkv + kv0) * hd, (nq + nkv + kv0 + pl["n_kv"]) * hd)]).to(wqkv.device)
    return wqkv[rows].contiguous(), wdense[:, q0 * hd:(q0 + pl["n_q"]) * hd].contiguous()


def shard_expert_scales(gu_scale, dn_scale, cfg, rank, world):
    """fp8 weight mode: the row scales that go with shard_experts' four tensors.  gu_scale [E + S, 2I], dn_scale [E + S, H] ->
    (routed gate/up [E / world, 2I], routed down [E / world, H], shared gate/up [2 * pad], shared down [H]).  A row of the shared
    down projection keeps ONE scale only while the rank's columns lie inside one pseudo-expert (world even for S = 2)."""
    E, S, I = cfg.num_experts, cfg.num_shared_experts or 0, cfg.moe_intermediate_size
    pl = shard_plan(cfg, world)
    e0 = rank * pl["n_experts"]
    gs, ds = gu_scale[e0:e0 + pl["n_experts"]].contiguous(), dn_scale[e0:e0 + pl["n_experts"]].contiguous()
    if not S:
        return gs, ds, None, None
    n, pad = pl["shared"], pl["shared_pad"]
    u0 = rank * n
    if u0 // I != (u0 + n - 1) // I:
        raise ValueError(f"fp8 weights: rank {rank}'s slice [{u0}, {u0 + n}) of the shared expert straddles two pseudo-experts of width {I}")
    sg = torch.cat([gu_scale[E + s, :I] for s in range(S)])         # scales of the shared gate_proj rows [S I]
    su = torch.cat([gu_scale[E + s, I:] for s in range(S)])
    ws_gs = torch.ones(2 * pad, dtype=torch.float32, device=gu_scale.device)
    ws_gs[:n], ws_gs[pad:pad + n] = sg[u0:u0 + n], su[u0:u0 + n]
    return gs, ds, ws_gs, dn_scale[E + u0 // I].contiguous()


def shard_experts(w_gate_up, w_down, cfg, rank, world):
    """Packed experts of a layer (bailing_moe.pack_experts: [E + S, 2I, H] / [E + S, H, I], shared pseudo-experts last) -> this rank's
    routed experts [E / world, ...] and its slice of the shared expert: ws_gate_up [2 * pad, H] (gate rows, up rows), ws_down [H, pad]
    with the slice zero-padded to a multiple of 64 units (silu(0) * 0 against zero down columns).  Pure indexing: bf16 weights and
    e4m3 bytes (uint8; zero byte = +0) alike."""
    E, S, I = cfg.num_experts, cfg.num_shared_experts or 0, cfg.moe_intermediate_size
    pl = shard_plan(cfg, world)
    e0 = rank * pl["n_experts"]
    gu, dn = w_gate_up[e0:e0 + pl["n_experts"]].contiguous(), w_down[e0:e0 + pl["n_experts"]].contiguous()
    if not S:
        return gu, dn, None, None
    H = w_gate_up.shape[2]
    sg = torch.cat([w_gate_up[E + s, :I] for s in range(S)], 0)        # shared gate_proj [S I, H]
    su = torch.cat([w_gate_up[E + s, I:] for s in range(S)], 0)
    sd = torch.cat([w_down[E + s] for s in range(S)], 1)               # shared down_proj [H, S I]
    n, pad = pl["shared"], pl["shared_pad"]
    u0 = rank * n
    ws_gu = torch.zeros(2 * pad, H, dtype=w_gate_up.dtype, device=w_gate_up.device)
    ws_gu[:n], ws_gu[pad:pad + n] = sg[u0:u0 + n], su[u0:u0 + n]
    ws_dn = torch.zeros(H, pad, dtype=w_down.dtype, device=w_down.device)
    ws_dn[:, :n] = sd[:, u0:u0 + n]
    return gu, dn, ws_gu, ws_dn


def nf4_shared_units(cfg, rank, world):
    """int4 (NF4) shards: this rank's units [u0, u0 + n) of the shared expert's S * I intermediate units.  NF4 quantises 64 consecutive
    k of a row with one absmax, and the down projection is sliced along k: the 64-unit blocks are dealt out whole — the first ranks
    take one more (16B-A3B, TP = 8: 44 blocks -> 6, 6, 6, 6, 5, 5, 5, 5) — so a shard holds the unsharded model's codes and absmax
    values themselves (no re-quantisation: the sharded model IS the int4 model)."""
    S, I = cfg.num_shared_experts or 0, cfg.moe_intermediate_size
    assert (S * I) % 64 == 0 and I % 64 == 0
    base, extra = divmod(S * I // 64, world)
    b0 = rank * base + min(rank, extra)
    return b0 * 64, (base + (1 if rank < extra else 0)) * 64


def shard_experts_nf4(w_gate_up, w_down, gu_absmax, dn_absmax, cfg, rank, world):
    """shard_experts + shard_expert_scales for NF4 experts: codes uint8 [E + S, 2I, H / 2] / [E + S, H, I / 2] (two per byte), absmax fp32
    [E + S, 2I, H / 64] / [E + S, H, I / 64] -> the rank's routed experts and its block-aligned slice of the shared expert, padded to
    `shared_pad` units with zero absmax (codes x 0 = 0).  Returns (gu, dn, ws_gu, ws_dn, gu_s, dn_s, ws_gu_s, ws_dn_s)."""
    E, S, I = cfg.num_experts, cfg.num_shared_experts or 0, cfg.moe_intermediate_size
    pl = shard_plan(cfg, world)
    e0, ne = rank * pl["n_experts"], pl["n_experts"]
    out = [w_gate_up[e0:e0 + ne].contiguous(), w_down[e0:e0 + ne].contiguous(), None, None,
           gu_absmax[e0:e0 + ne].contiguous(), dn_absmax[e0:e0 + ne].contiguous(), None, None]
    if not S:
        return tuple(out)
    u0, n = nf4_shared_units(cfg, rank, world)
    pad = pl["shared_pad"]
    assert n <= pad
    dev = w_gate_up.device
    for base, q, rows_per_unit in ((2, w_gate_up, None), (6, gu_absmax, None)):      # gate / up ROWS of the shared Linear: whole rows travel
        sg = torch.cat([q[E + s, :I] for s in range(S)], 0)
        su = torch.cat([q[E + s, I:] for s in range(S)], 0)
        t = torch.zeros((2 * pad,) + tuple(sg.shape[1:]), dtype=q.dtype, device=dev)
        t[:n], t[pad:pad + n] = sg[u0:u0 + n], su[u0:u0 + n]
        out[base] = t
    sd = torch.cat([w_down[E + s] for s in range(S)], 1)               # codes of the shared down_proj [H, S I / 2]
    sa = torch.cat([dn_absmax[E + s] for s in range(S)], 1)            # its absmax [H, S I / 64]
    ws_dn = torch.zeros(sd.shape[0], pad // 2, dtype=sd.dtype, device=dev)
    ws_dn[:, :n // 2] = sd[:, u0 // 2:(u0 + n) // 2]
    ws_da = torch.zeros(sa.shape[0], pad // 64, dtype=sa.dtype, device=dev)
    ws_da[:, :n // 64] = sa[:, u0 // 64:(u0 + n) // 64]
    out[3], out[7] = ws_dn, ws_da
    return tuple(out)


def shard_rf_block(w12, b12, w3, rank, world):
    """RF ResBlock MLP: w12 [2 hid, w] (gate rows, up rows) -> [2 hid / world, w]; b12 likewise; w3 [w, hid] -> [w, hid / world]."""
    hid = w12.shape[0] // 2
    n = hid // world
    u0 = rank * n
    return (torch.cat((w12[u0:u0 + n], w12[hid + u0:hid + u0 + n]), 0).contiguous(),
            torch.cat((b12[u0:u0 + n], b12[hid + u0:hid + u0 + n]), 0).contiguous(), w3[:, u0:u0 + n].contiguous())


# ------------------------------------------------------------------------------------------------------------------------
# communicator
# ------------------------------------------------------------------------------------------------------------------------
class TpCommunicator:
    """One rank's view of the inboxes / flags of all ranks (mn_tp_comm)."""

    def __init__(self, rank, world, inbox_ptrs, flag_ptrs, cap, rows_cap, err, keep=()):
        self.rank, self.world, self.cap, self.rows_cap = rank, world, cap, rows_cap
        self._inbox = (C.c_void_p * world)(*inbox_ptrs)
        self._flags = (C.c_void_p * world)(*flag_ptrs)
        self.err = err
        self._keep = keep
        s = TpComm()
        s.rank, s.world, s.cap, s.rows_cap, s.epoch = rank, world, cap, rows_cap, 0
        s.inbox, s.flags = C.cast(self._inbox, _lib.PP), C.cast(self._flags, _lib.PP)
        s.err = err.data_ptr()
        self.struct = s

    @staticmethod
    def simulated(world, rows_cap, width, device="cuda"):
        """`world` communicators in ONE process on ONE device (TpSimGroup): the peers' buffers are ordinary allocations."""
        cap = rows_cap * width
        inbox = [torch.zeros(2 * world * cap, dtype=torch.float32, device=device) for _ in range(world)]
        flags = [torch.zeros(world * rows_cap, dtype=torch.int32, device=device) for _ in range(world)]
        errs = [torch.zeros(1, dtype=torch.int32, device=device) for _ in range(world)]
        return [TpCommunicator(r, world, [t.data_ptr() for t in inbox], [t.data_ptr() for t in flags], cap, rows_cap, errs[r],
                               keep=(inbox, flags)) for r in range(world)]

    @staticmethod
    def from_process_group(dist, rows_cap, width, device):
        """One process per GPU (torch.distributed initialised; backend "nccl" = RCCL): allocate this rank's fine-grained inbox and
        flags, exchange their IPC handles (all_gather_object: setup traffic goes through the process group) and map the peers'."""
        rank, world = dist.get_rank(), dist.get_world_size()
        cap = rows_cap * width
        L = lib()
        mine = []
        for nbytes in (2 * world * cap * 4, world * rows_cap * 4):
            p = C.c_void_p()
            check(L.mn_tp_alloc(nbytes, C.byref(p)), "mn_tp_alloc")
            h = (C.c_char * 64)()
            check(L.mn_tp_ipc_handle(p, h), "mn_tp_ipc_handle")
            mine.append((p.value, bytes(h)))
        handles = [None] * world
        dist.all_gather_object(handles, [h for _, h in mine])
        inbox_ptrs, flag_ptrs = [], []
        for r in range(world):
            for k, out in ((0, inbox_ptrs), (1, flag_ptrs)):
                if r == rank:
                    out.append(mine[k][0])
                else:
                    p = C.c_void_p()
                    check(L.mn_tp_ipc_open(C.create_string_buffer(handles[r][k], 64), C.byref(p)), "mn_tp_ipc_open")
                    out.append(p.value)
        dist.barrier()
        err = torch.zeros(1, dtype=torch.int32, device=device)
        c = TpCommunicator(rank, world, inbox_ptrs, flag_ptrs, cap, rows_cap, err)
        c._ipc = (dist, [p for r, p in enumerate(inbox_ptrs + flag_ptrs) if r % world != rank], [p for p, _ in mine])
        return c

    def close(self):
        """Unmap the peers' buffers and free this rank's (communicators made by from_process_group; a collective: every rank calls
        it, after its last composite has completed)."""
        ipc = getattr(self, "_ipc", None)
        if ipc is None:
            return
        dist, mapped, own = ipc
        torch.cuda.synchronize()
        dist.barrier()                                   # nobody unmaps while a peer may still push
        L = lib()
        for p in mapped:
            L.mn_tp_ipc_close(C.c_void_p(p))
        dist.barrier()
        for p in own:
            L.mn_tp_free(C.c_void_p(p))
        self._ipc = None

    @staticmethod
    def relayed(dist, rows_cap, width, device):
        """RCCL fallback transport (no peer mapping: e.g. GPUs without xGMI peer access, or debugging): every push lands in LOCAL
        memory (this rank's own inbox slab; the peers' copies go to a sink), and between two segments the host enqueues an
        all-gather of the slabs over the process group plus a flag fill (`relay`).  Same kernels and segment mechanism; one
        collective per all-reduce instead of posted xGMI writes."""
        rank, world = dist.get_rank(), dist.get_world_size()
        cap = rows_cap * width
        inbox = torch.zeros(2, world, cap, dtype=torch.float32, device=device)
        sink = torch.zeros(2, world, cap, dtype=torch.float32, device=device)
        flags = torch.zeros(world * rows_cap, dtype=torch.int32, device=device)
        sink_flags = torch.zeros(world * rows_cap, dtype=torch.int32, device=device)
        err = torch.zeros(1, dtype=torch.int32, device=device)
        c = TpCommunicator(rank, world, [inbox.data_ptr() if r == rank else sink.data_ptr() for r in range(world)],
                           [flags.data_ptr() if r == rank else sink_flags.data_ptr() for r in range(world)], cap, rows_cap, err,
                           keep=(sink, sink_flags))
        c._relay = (dist, inbox, flags)
        c.struct.two_shot_rows = -1              # the relay delivers whole rows between segments: one-shot form only
        return c

    def relay(self, epoch, n_floats):
        """Relayed transport only: deliver all-reduce `epoch` (every rank's first n_floats of its slab) and raise the arrival flags."""
        dist, inbox, flags = self._relay
        par = epoch & 1
        dist.all_gather([inbox[par, r, :n_floats] for r in range(self.world)], inbox[par, self.rank, :n_floats])
        flags.fill_(int(epoch))

    def check_err(self):
        e = int(self.err.item())
        if e:
            raise RuntimeError(f"tensor-parallel all-reduce: rank {self.rank} gave up waiting for sender {e & 0xff} (a peer died or the "
                               "ranks' launch sequences diverged)")

    PUSH, REDUCE, GATHER = 1, 2, 4

    def allreduce_segments(self, rows, width):
        """1 (one-shot) or 2 (two-shot: above `two_shot_rows` rows) — segments / epochs an all-reduce of rows x width takes here."""
        return int(lib().mn_tp_allreduce_segments(C.byref(self.struct), rows, width))

    def all_reduce(self, x, out=None, phase=7):
        """out = sum over ranks of x (fp32 [M, D]) through mn_allreduce_oneshot; phase = PUSH, GATHER (two-shot only: the owners'
        reduce + all-gather push, a no-op for a one-shot all-reduce), REDUCE, or all of them (7)."""
        assert x.dtype == torch.float32 and x.is_cuda and x.is_contiguous() and x.dim() == 2
        M, D = x.shape
        if out is None and phase & self.REDUCE:
            out = torch.empty_like(x)
        check(lib().mn_allreduce_oneshot(C.byref(self.struct), ptr(x), D, ptr(out), 0 if out is None else out.stride(0), M, D, phase,
                                         current_stream()), "mn_allreduce_oneshot")
        return out


# ------------------------------------------------------------------------------------------------------------------------
# shards
# ------------------------------------------------------------------------------------------------------------------------
class TpDecoderShard:
    """One rank's share of the decoder stack: its heads, its KV heads' arena, its experts, its slice of the shared expert."""

    def __init__(self, dec, rank, world):
        """dec: a full BailingMoeDecoder (packed experts); its weights are sliced (copies), its replicated tensors shared."""
        from .bailing_moe import rope_tables  # noqa: F401  (tables are shared with `dec`)
        cfg = dec.cfg
        self.cfg, self.rank, self.world, self.device = cfg, rank, world, dec.device
        pl = shard_plan(cfg, world)
        self.plan = pl
        self.t_max, self.n_seq = dec.t_max, dec.n_seq
        L = cfg.num_hidden_layers
        self.layers = getattr(self, "_shard_layers", None) or []
        self.weights = getattr(dec, "weights", "bf16")
        if self.weights == "int4" and getattr(dec, "stream_fmt", "int4") != "int4":
            raise NotImplementedError("tensor parallel shards of an int4 model whose widths are not multiples of 64 (its experts run as bf16 values)")
        for ly in (dec.layers or []):
            self.layers.append(self._shard_layer(ly, cfg, rank, world))
        hd = cfg.head_dim
        self.kv_cache = torch.zeros(L, self.n_seq, 2, pl["n_kv"], self.t_max, hd, dtype=torch.float32, device=self.device)
        keys = ("ln1", "wqkv", "wdense", "ln2", "gate", "image_gate", "w_gate_up", "w_down", "ws_gate_up", "ws_down")
        skeys = ("w_gate_up_scale", "w_down_scale", "ws_gate_up_scale", "ws_down_scale")
        self._arrays = {k: ptr_array([ly.get(k) for ly in self.layers]) for k in keys + skeys}
        s = Llm()
        s.hidden, s.n_layers, s.n_q, s.n_kv, s.head_dim = cfg.hidden_size, L, pl["n_q"], pl["n_kv"], hd
        s.n_experts, s.top_k, s.n_shared_slots = cfg.num_experts, cfg.num_experts_per_tok, 0
        s.moe_inter, s.norm_topk_prob, s.rms_eps = cfg.moe_intermediate_size, int(cfg.norm_topk_prob), cfg.rms_norm_eps
        for k in keys[:8]:
            setattr(s, k, C.cast(self._arrays[k], _lib.PP))
        if not cfg.multi_gate:
            s.image_gate = None
        s.final_norm = ptr(dec.final_norm)
        s.cos_tab, s.sin_tab, s.n_pos = ptr(dec.cos), ptr(dec.sin), dec.cos.shape[0]
        if dec.mrope_section is not None:
            s.mrope_sec_t, s.mrope_sec_h = dec.mrope_section[0], dec.mrope_section[1]
        self.mrope_section = dec.mrope_section
        s.wfmt = _lib.WFMT[self.weights]
        if self.weights in _lib.W8:
            s.w_gate_up_scale = C.cast(self._arrays["w_gate_up_scale"], _lib.PP)
            s.w_down_scale = C.cast(self._arrays["w_down_scale"], _lib.PP)
        self.struct = s
        t = LlmTp()
        t.expert0, t.n_local_experts, t.shared_inter = rank * pl["n_experts"], pl["n_experts"], pl["shared_pad"]
        if pl["shared_pad"]:
            t.ws_gate_up, t.ws_down = C.cast(self._arrays["ws_gate_up"], _lib.PP), C.cast(self._arrays["ws_down"], _lib.PP)
            if self.weights in _lib.W8:
                t.ws_gate_up_scale = C.cast(self._arrays["ws_gate_up_scale"], _lib.PP)
                t.ws_down_scale = C.cast(self._arrays["ws_down_scale"], _lib.PP)
        self.tp = t
        self._keep = (dec.final_norm, dec.cos, dec.sin)
        self._ws = {}

    @staticmethod
    def _shard_layer(ly, cfg, rank, world):
        """One layer of a full decoder (packed experts; bf16, or e4m3 bytes + row scales) -> this rank's tensors."""
        wqkv, wdense = shard_attention(ly["wqkv"], ly["wdense"], cfg, rank, world)
        if ly["w_gate_up"].dtype == torch.uint8 and ly["w_gate_up_scale"].dim() == 3:      # NF4: codes + one absmax per 64 k
            gu, dn, wsg, wsd, gs, ds, wsgs, wsds = shard_experts_nf4(ly["w_gate_up"], ly["w_down"], ly["w_gate_up_scale"], ly["w_down_scale"],
                                                                    cfg, rank, world)
            return dict(ln1=ly["ln1"], wqkv=wqkv, wdense=wdense, ln2=ly["ln2"], gate=ly["gate"], image_gate=ly.get("image_gate"),
                        w_gate_up=gu, w_down=dn, ws_gate_up=wsg, ws_down=wsd, w_gate_up_scale=gs, w_down_scale=ds,
                        ws_gate_up_scale=wsgs, ws_down_scale=wsds)
        gu, dn, wsg, wsd = shard_experts(ly["w_gate_up"], ly["w_down"], cfg, rank, world)
        out = dict(ln1=ly["ln1"], wqkv=wqkv, wdense=wdense, ln2=ly["ln2"], gate=ly["gate"], image_gate=ly.get("image_gate"),
                   w_gate_up=gu, w_down=dn, ws_gate_up=wsg, ws_down=wsd)
        if ly["w_gate_up"].dtype == torch.uint8:
            gs, ds, wsgs, wsds = shard_expert_scales(ly["w_gate_up_scale"], ly["w_down_scale"], cfg, rank, world)
            out.update(w_gate_up_scale=gs, w_down_scale=ds, ws_gate_up_scale=wsgs, ws_down_scale=wsds)
        return out

    @classmethod
    def synthetic(cls, cfg, device, rank, world, seed=0, t_max=2048, n_seq=3, n_pos=None, weights="bf16"):
        """A rank's shard of the random-init 16B-A3B-style stack WITHOUT ever holding the full model: every layer is synthesised
        with the reference's parameter names (same seed on every rank -> the same full weights), packed, sliced and dropped.
        weights="fp8": the packed experts of each layer are quantised (as the unsharded model's) before slicing."""
        from .bailing_moe import BailingMoeDecoder, pack_experts, rope_tables
        from .configuration import llm_layer_param_shapes
        from .synth import synth_tensor

        class _Full:                                    # the few attributes __init__ reads from a full decoder
            pass
        full = _Full()
        full.cfg, full.device, full.t_max, full.n_seq = cfg, torch.device(device), t_max, n_seq
        hd = cfg.head_dim
        full.cos, full.sin = rope_tables(hd, cfg.rope_theta, n_pos or t_max, device)
        full.final_norm = synth_tensor("model.norm.weight", (cfg.hidden_size,), seed, device, torch.bfloat16)
        full.mrope_section = [16, 24, 24] if cfg.rope_scaling is not None else None
        self = cls.__new__(cls)
        self._shard_layers = []
        for li in range(cfg.num_hidden_layers):
            sd = {k: synth_tensor(k, v, seed, device, torch.bfloat16) for k, v in llm_layer_param_shapes(cfg, li).items()}
            p = f"model.layers.{li}"
            gu, dn = pack_experts(sd, p + ".mlp", cfg)
            ly = dict(ln1=sd[p + ".input_layernorm.weight"], wqkv=sd[p + ".attention.query_key_value.weight"],
                      wdense=sd[p + ".attention.dense.weight"], ln2=sd[p + ".post_attention_layernorm.weight"],
                      gate=sd[p + ".mlp.gate.weight"], image_gate=sd.get(p + ".mlp.image_gate.weight") if cfg.multi_gate else None,
                      w_gate_up=gu, w_down=dn)
            if weights in _lib.W8:
                from .bailing_moe import quantize_layer_experts
                quantize_layer_experts(ly, weights, cfg.num_shared_experts or 0)
            self._shard_layers.append(cls._shard_layer(ly, cfg, rank, world))
            del sd, gu, dn, ly
        full.layers = None
        full.weights = weights
        self.__init__(full, rank, world)
        return self

    def n_segments(self, comm=None, rows=1):
        """Segments of one step: 2 L + 1 with one-shot all-reduces; a two-shot all-reduce (above comm.two_shot_rows rows) adds the
        owners' reduce + all-gather segment."""
        base = int(lib().mn_llm_tp_segments(C.byref(self.struct)))
        ars = 1 if comm is None else int(lib().mn_tp_allreduce_segments(C.byref(comm.struct), rows, self.cfg.hidden_size))
        return (base - 1) * ars + 1

    def weight_bytes(self):
        return sum(t.numel() * t.element_size() for ly in self.layers for k, t in ly.items() if t is not None and k not in ("ln1", "ln2"))

    def step_tp(self, comm, x, row_seq, row_slot, row_pos, row_len, key_mask=None, image_mask=None, out=None, rows=None, x_row_div=1,
                seg_begin=0, seg_end=None):
        """mn_llm_step_tp on this shard (arguments as BailingMoeDecoder.step); [seg_begin, seg_end) default = all segments."""
        M = rows or x.shape[0]
        ldx = 0 if (rows is not None and x.shape[0] == 1) else x.stride(0)
        if self.mrope_section is not None and row_pos.dim() == 1:
            row_pos = row_pos[:M].unsqueeze(0).expand(3, M).contiguous()
        if out is None:
            out = torch.empty(M, self.cfg.hidden_size, dtype=torch.float32, device=self.device)
        key = (M, torch.cuda.current_stream().cuda_stream)
        if key not in self._ws:
            n = lib().mn_llm_tp_workspace_bytes(C.byref(self.struct), C.byref(self.tp), M, self.t_max)
            self._ws[key] = torch.empty(n, dtype=torch.uint8, device=self.device)
        ws = self._ws[key]
        n_seg = self.n_segments(comm, M)
        check(lib().mn_llm_step_tp(C.byref(self.struct), C.byref(self.tp), C.byref(comm.struct), ptr(x), ldx, x_row_div, M, ptr(image_mask),
                                   ptr(row_seq), ptr(row_slot), ptr(row_pos), ptr(row_len), ptr(key_mask),
                                   0 if key_mask is None else key_mask.stride(0), ptr(self.kv_cache), self.n_seq, self.t_max, ptr(out),
                                   ptr(ws), ws.numel(), seg_begin, n_seg if seg_end is None else seg_end, current_stream()),
              "mn_llm_step_tp")
        return out


class TpRfShard:
    """One rank's share of the RF head: its slice of every ResBlock's SwiGLU width; everything else shared with the full head."""

    def __init__(self, rf, rank, world):
        self.rf, self.rank, self.world = rf, rank, world
        assert rf.hidden % world == 0 and (rf.hidden // world) % 64 == 0, "RF SwiGLU width must split into multiples of 64"
        self.hidden = rf.hidden // world
        self.weights = getattr(rf, "weights", "bf16")
        nf4 = self.weights == "int4"
        if nf4 and rf.lists["w12"][0].dtype != torch.uint8:
            raise NotImplementedError("tensor parallel shards of an int4 RF head whose widths are not multiples of 64 (it runs as bf16 values)")
        w12, b12, w3 = [], [], []
        for b in range(rf.depth):
            if nf4:      # codes, two per byte: w12's rows travel whole; w3 is sliced along k in bytes (hidden / world is a multiple of 64: whole absmax blocks)
                n, u0, hid = self.hidden, rank * self.hidden, rf.hidden
                q12, bq = rf.lists["w12"][b], rf.lists["b12"][b]
                a = torch.cat((q12[u0:u0 + n], q12[hid + u0:hid + u0 + n]), 0).contiguous()
                bb = torch.cat((bq[u0:u0 + n], bq[hid + u0:hid + u0 + n]), 0).contiguous()
                c = rf.lists["w3"][b][:, u0 // 2:(u0 + n) // 2].contiguous()
            else:
                a, bb, c = shard_rf_block(rf.lists["w12"][b], rf.lists["b12"][b], rf.lists["w3"][b], rank, world)
            w12.append(a); b12.append(bb); w3.append(c)
        self.lists = dict(rf.lists, w12=w12, b12=b12, w3=w3)
        self._arrays = {k: ptr_array(v) for k, v in self.lists.items()}
        if self.weights in _lib.W8:      # row scales: w12's rows are sliced like its weights, w3's columns share the full rows' scales
            hid, n = rf.hidden, rf.hidden // world
            u0 = rank * n
            self.scales = dict(w12=[torch.cat((sc[u0:u0 + n], sc[hid + u0:hid + u0 + n])).contiguous() for sc in rf.scales["w12"]],
                               w3=[sc[:, u0 // 64:(u0 + n) // 64].contiguous() for sc in rf.scales["w3"]] if nf4 else rf.scales["w3"])
            self._scale_arrays = {k: ptr_array(v) for k, v in self.scales.items()}
        s = RfHead()
        s.w, s.depth, s.hidden, s.z_dim, s.target, s.steps, s.llm_hidden = rf.w, rf.depth, self.hidden, rf.w, rf.target, rf.steps, rf.llm_hidden
        for k in ("vis_w", "vis_b", "vis_ln_g", "vis_ln_b", "cond_w", "cond_b", "in_w", "in_b", "temb", "ada_w", "ada_b", "fin_w", "fin_b"):
            setattr(s, k, ptr(rf.t[k]))
        for k, arr in self._arrays.items():
            setattr(s, k, C.cast(arr, _lib.PP))
        s.wfmt = _lib.WFMT[self.weights]
        if self.weights in _lib.W8:
            s.w12_scale = C.cast(self._scale_arrays["w12"], _lib.PP)
            s.w3_scale = C.cast(self._scale_arrays["w3"], _lib.PP)
        self.struct = s
        self.target = rf.target
        self._ws = {}

    def n_segments(self, comm=None, rows=1):
        base = int(lib().mn_rf_tp_segments(C.byref(self.struct)))
        ars = 1 if comm is None else int(lib().mn_tp_allreduce_segments(C.byref(comm.struct), rows, self.rf.w))
        return (base - 1) * ars + 1

    def sample_tp(self, comm, hidden, noise, temperature=1.0, text_cfg=3.0, image_cfg=1.1, out=None, n_images=1, seg_begin=0, seg_end=None):
        rows = hidden.shape[0]
        key = (rows, torch.cuda.current_stream().cuda_stream)
        if key not in self._ws:
            n = lib().mn_rf_tp_workspace_bytes(C.byref(self.struct), rows)
            self._ws[key] = torch.empty(n, dtype=torch.uint8, device=hidden.device)
        ws = self._ws[key]
        if out is None:
            out = torch.empty(noise.shape, dtype=torch.float32, device=hidden.device)
        n_seg = self.n_segments(comm, rows)
        check(lib().mn_rf_sample_tp(C.byref(self.struct), C.byref(comm.struct), ptr(hidden), hidden.stride(0), rows, n_images, ptr(noise),
                                    float(temperature), float(text_cfg), float(image_cfg), ptr(out), ptr(ws), ws.numel(), seg_begin,
                                    n_seg if seg_end is None else seg_end, current_stream()), "mn_rf_sample_tp")
        return out


def ops_lmhead(hidden, lm_slice, vocab0):
    from . import ops
    idx, val = ops.lmhead_argmax(hidden.contiguous(), lm_slice, vocab_offset=vocab0)
    return idx, val


def vocab_parallel_pick(dist, pair):
    """(idx int64 [M], val fp32 [M]) of this rank's vocabulary slice -> the global greedy ids [M]: the pair with the largest logit,
    the lowest id among equals (selection over `world` candidates per row; the logits themselves come from mn_lmhead_argmax)."""
    idx, val = pair
    world = dist.get_world_size()
    idxs = [torch.empty_like(idx) for _ in range(world)]
    vals = [torch.empty_like(val) for _ in range(world)]
    dist.all_gather(idxs, idx)
    dist.all_gather(vals, val)
    return pick_best(torch.stack(idxs), torch.stack(vals))


def pick_best(idxs, vals):
    """idxs / vals [world, M] -> [M]: max logit, ties -> lowest id; NaN counts as the maximum (torch.argmax's order), so a row of
    NaN logits still yields a valid id."""
    vals = torch.nan_to_num(vals, nan=float("inf"), posinf=float("inf"))
    best = vals.max(dim=0).values
    cand = torch.where(vals == best.unsqueeze(0), idxs, torch.full_like(idxs, torch.iinfo(torch.int64).max))
    return cand.min(dim=0).values


# ------------------------------------------------------------------------------------------------------------------------
# one rank of a real TP group / all ranks simulated on one GPU — both expose the decoder + sampler interface of generate_images
# ------------------------------------------------------------------------------------------------------------------------
class _TpDecoderBase:
    """The slice of BailingMoeDecoder's interface that generate_images / generate use."""
    MAX_ROWS = 2048
    single_stream = True        # generate_images(n_groups > 1) is refused: one communicator = one stream (two-parity inbox)

    def _row_cap(self):
        """fp8 weights are served by the <= 64-row streaming kernels only (mingnative.h section 7)."""
        fp8 = any(getattr(self, a, "bf16") in _lib.W8 for a in ("weights", "rf_weights"))      # an 8-bit weight mode (fp8 or int8)
        return min(64 if fp8 else self.MAX_ROWS, self.rows_cap)

    def max_rows(self):
        return self._row_cap()

    def embed(self, ids):
        return self.full.embed(ids)

    def logits(self, hidden):
        return self.full.logits(hidden)            # lm_head replicated (vocabulary split: next step)

    def greedy(self, hidden):
        return self.full.greedy(hidden)

    def sample(self, hidden, u, temperature=1.0, top_k=50, top_p=1.0):
        return self.full.sample(hidden, u, temperature, top_k, top_p)      # lm_head replicated: every rank draws the same token from the same u

    def prefill_many(self, embeds, seqs, past=0):
        """B prompts of equal length in lock-step (BailingMoeDecoder.prefill_many): embeds fp32 [B, T, H] -> hidden [B, T, H]."""
        B, T, H = embeds.shape
        assert past + T <= self.t_max and len(seqs) == B
        dev = self.device
        x = embeds.reshape(B * T, H).contiguous()
        seq = torch.tensor(list(seqs), dtype=torch.int32, device=dev).repeat_interleave(T)
        slot = (torch.arange(T, dtype=torch.int32, device=dev) + past).repeat(B)
        out = torch.empty(B * T, H, dtype=torch.float32, device=dev)
        step = self.max_rows()
        for r0 in range(0, B * T, step):
            r1 = min(B * T, r0 + step)
            sl = slot[r0:r1].contiguous()
            self.step(x[r0:r1], seq[r0:r1].contiguous(), sl, sl, (sl + 1).contiguous(), None, None, out=out[r0:r1])
        return out.reshape(B, T, H)

    def prefill(self, embeds, seq=0, past=0, image_mask=None, chunk=None):
        T = embeds.shape[0]
        assert past + T <= self.t_max
        step = min(self.max_rows(), chunk or self.max_rows())
        outs = []
        for c0 in range(0, T, step):
            m = min(step, T - c0)
            slot = torch.arange(past + c0, past + c0 + m, dtype=torch.int32, device=self.device)
            seqs = torch.full((m,), seq, dtype=torch.int32, device=self.device)
            im = None if image_mask is None else image_mask[c0:c0 + m].to(self.device, torch.uint8).contiguous()
            outs.append(self.step(embeds[c0:c0 + m].contiguous(), seqs, slot, slot, slot + 1, None, im))
        return torch.cat(outs, 0)


    # the wrapper's long-prompt entry points (MingUniVisionForConditionalGeneration.generate): a TP group prefills long prompts in
    # passes of max_rows() rows through the same sharded step — fp32-class numerics (hi/lo operands) at every row count
    def prefill_wide(self, embeds, seq=0, past=0, image_mask=None):
        return self.prefill(embeds, seq=seq, past=past, image_mask=image_mask)

    def prefill_mfma(self, embeds, seq=0, past=0, image_mask=None):
        return self.prefill(embeds, seq=seq, past=past, image_mask=image_mask)[-1:]


class TpSimGroup(_TpDecoderBase):
    """All `world` ranks of a TP group in one process on one GPU: every shard has its own weights, KV arena, workspace and
    communicator; a composite runs segment k on every rank before segment k + 1 on any (one stream), so each all-reduce's pushes
    precede its consumers in stream order and the flag waits are satisfied on arrival.  Same kernels, same flag protocol as one
    process per GPU — what cannot be exercised here is the cross-device memory path (unmeasured on hardware)."""

    def __init__(self, dec, rf, world, rows_cap=64):
        self.full, self.rf_full, self.world, self.rows_cap = dec, rf, world, rows_cap
        self.cfg, self.device, self.t_max, self.n_seq = dec.cfg, dec.device, dec.t_max, dec.n_seq
        width = max(dec.cfg.hidden_size, rf.w if rf is not None else 0)
        self.comms = TpCommunicator.simulated(world, rows_cap, width, dec.device)
        self.shards = [TpDecoderShard(dec, r, world) for r in range(world)]
        self.rf_shards = [TpRfShard(rf, r, world) for r in range(world)] if rf is not None else None
        self.target = rf.target if rf is not None else None
        self.weights, self.rf_weights = getattr(dec, "weights", "bf16"), getattr(rf, "weights", "bf16")

    # -- decoder -------------------------------------------------------------------------------------------------
    def step(self, x, row_seq, row_slot, row_pos, row_len, key_mask=None, image_mask=None, out=None, rows=None, x_row_div=1,
             distinct_sequences=False):      # (the TP composites keep the stand-alone append: accepted for interface parity)
        M = rows or x.shape[0]
        outs = [out if r == 0 and out is not None else torch.empty(M, self.cfg.hidden_size, dtype=torch.float32, device=self.device)
                for r in range(self.world)]
        for seg in range(self.shards[0].n_segments(self.comms[0], M)):
            for r, sh in enumerate(self.shards):
                sh.step_tp(self.comms[r], x, row_seq, row_slot, row_pos, row_len, key_mask, image_mask, out=outs[r], rows=rows,
                           x_row_div=x_row_div, seg_begin=seg, seg_end=seg + 1)
        self.last_rank_outputs = outs
        return outs[0]

    def copy_sequence(self, src, dst, n):
        for sh in self.shards:
            sh.kv_cache[:, dst, :, :, :n].copy_(sh.kv_cache[:, src, :, :, :n])

    def sample(self, hidden, u, temperature=1.0, top_k=50, top_p=1.0):
        return self.full.sample(hidden, u, temperature, top_k, top_p)

    def greedy(self, hidden):
        """The vocabulary-parallel greedy pick of TpRank.greedy with every rank's slice computed in this process."""
        lm = self.full.lm_head
        V = lm.shape[0]
        n = -(-V // self.world)
        pairs = [ops_lmhead(hidden, lm[min(V, r * n):min(V, (r + 1) * n)], min(V, r * n)) for r in range(self.world) if r * n < V]
        return pick_best(torch.stack([p[0] for p in pairs]), torch.stack([p[1] for p in pairs]))

    # -- RF sampler ----------------------------------------------------------------------------------------------
    def rf_max_rows(self):
        return self._row_cap()

    def rf_sample(self, hidden, noise, temperature=1.0, text_cfg=3.0, image_cfg=1.1, out=None, n_images=1):
        """RectifiedFlowLoss.sample through the group (the TOKEN sampler above keeps the decoder interface's name `sample`)."""
        outs = [out if r == 0 and out is not None else torch.empty(noise.shape, dtype=torch.float32, device=hidden.device)
                for r in range(self.world)]
        for seg in range(self.rf_shards[0].n_segments(self.comms[0], hidden.shape[0])):
            for r, sh in enumerate(self.rf_shards):
                sh.sample_tp(self.comms[r], hidden, noise, temperature, text_cfg, image_cfg, out=outs[r], n_images=n_images,
                             seg_begin=seg, seg_end=seg + 1)
        self.last_rank_latents = outs
        return outs[0]

    def check_err(self):
        for c in self.comms:
            c.check_err()

    def sampler(self):
        """The RF-head object generate_images expects (sample / max_rows / target)."""
        return _SamplerView(self)


class _SamplerView:
    def __init__(self, grp):
        self._g, self.target = grp, grp.target

    def max_rows(self):
        return self._g.rf_max_rows()

    def sample(self, *a, **kw):
        return self._g.rf_sample(*a, **kw)


class TpRank(_TpDecoderBase):
    """One rank of a real TP group (one process per GPU): the same interface.  transport "xgmi": peer-mapped inboxes, a composite is
    ONE call (all segments, the waits are real); "rccl": TpCommunicator.relayed, segment by segment with an all-gather in between."""

    def __init__(self, dec, rf, dist, rows_cap=64, transport="xgmi", shard=None, embed_fn=None, logits_fn=None):
        """dec: the full decoder (its weights are sliced) or None with a prebuilt `shard` (TpDecoderShard.synthetic) plus the
        replicated embed / logits callables."""
        self.full, self.rf_full, self.rows_cap, self.transport = dec, rf, rows_cap, transport
        self._dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.shard = shard if shard is not None else TpDecoderShard(dec, self.rank, self.world)
        src = dec if dec is not None else self.shard
        self.cfg, self.device, self.t_max, self.n_seq = src.cfg, src.device, src.t_max, src.n_seq
        self._embed, self._logits = embed_fn, logits_fn
        width = max(self.cfg.hidden_size, rf.w if rf is not None else 0)
        make = TpCommunicator.from_process_group if transport == "xgmi" else TpCommunicator.relayed
        self.comm = make(dist, rows_cap, width, self.device)
        self.rf_shard = TpRfShard(rf, self.rank, self.world) if rf is not None else None
        self.target = rf.target if rf is not None else None
        self.weights, self.rf_weights = self.shard.weights, getattr(rf, "weights", "bf16")

    def embed(self, ids):
        return self._embed(ids) if self._embed is not None else self.full.embed(ids)

    def logits(self, hidden):
        return self._logits(hidden) if self._logits is not None else self.full.logits(hidden)

    def sample(self, hidden, u, temperature=1.0, top_k=50, top_p=1.0):
        """Sampled pick: the warped distribution needs the whole row of logits, so every rank computes it from the replicated lm_head
        and draws at the same uniform (the callers seed their generators alike) — identical tokens on all ranks, no exchange."""
        if self.full is None or self.full.lm_head is None:
            raise RuntimeError("TpRank.sample needs the lm_head (build the rank from a full decoder with vocabulary weights)")
        ids = self.full.sample(hidden, u, temperature, top_k, top_p)
        self.check_err()
        return ids

    def greedy(self, hidden):
        """Greedy pick with the lm_head split over the vocabulary (each rank streams V / world rows of it): mn_lmhead_argmax on the
        rank's slice with its vocabulary offset, then the ranks' (logit, id) pairs are gathered over the process group and the
        best pair wins — ties to the lowest id, like the unsharded arg-max."""
        if self.full is None or self.full.lm_head is None:
            raise RuntimeError("TpRank.greedy needs the lm_head (build the rank from a full decoder with vocabulary weights)")
        lm = self.full.lm_head
        V = lm.shape[0]
        n = -(-V // self.world)
        v0, v1 = min(V, self.rank * n), min(V, (self.rank + 1) * n)
        ids = vocab_parallel_pick(self._dist, ops_lmhead(hidden, lm[v0:v1], v0))
        self.check_err()                                 # a host sync point anyway (the caller reads the ids): expired waits surface here
        return ids

    def _segmented(self, call, n_seg, n_floats):
        """Relayed transport: run the composite one segment at a time, delivering each all-reduce through the process group."""
        base = self.comm.struct.epoch
        out = None
        for seg in range(n_seg):
            out = call(seg, seg + 1)
            if seg + 1 < n_seg:
                self.comm.relay(base + seg + 1, n_floats)
        return out

    def step(self, x, row_seq, row_slot, row_pos, row_len, key_mask=None, image_mask=None, out=None, rows=None, x_row_div=1,
             distinct_sequences=False):      # (the TP composites keep the stand-alone append: accepted for interface parity)
        M = rows or x.shape[0]
        if out is None:
            out = torch.empty(M, self.cfg.hidden_size, dtype=torch.float32, device=self.device)

        def call(s0=0, s1=None):
            return self.shard.step_tp(self.comm, x, row_seq, row_slot, row_pos, row_len, key_mask, image_mask, out=out, rows=rows,
                                      x_row_div=x_row_div, seg_begin=s0, seg_end=s1)
        if self.transport == "xgmi":
            return call()
        return self._segmented(call, self.shard.n_segments(self.comm, M), M * self.cfg.hidden_size)

    def copy_sequence(self, src, dst, n):
        self.shard.kv_cache[:, dst, :, :, :n].copy_(self.shard.kv_cache[:, src, :, :, :n])

    def rf_max_rows(self):
        return self._row_cap()

    def rf_sample(self, hidden, noise, temperature=1.0, text_cfg=3.0, image_cfg=1.1, out=None, n_images=1):
        if out is None:
            out = torch.empty(noise.shape, dtype=torch.float32, device=hidden.device)

        def call(s0=0, s1=None):
            return self.rf_shard.sample_tp(self.comm, hidden, noise, temperature, text_cfg, image_cfg, out=out, n_images=n_images,
                                           seg_begin=s0, seg_end=s1)
        if self.transport == "xgmi":
            return call()
        return self._segmented(call, self.rf_shard.n_segments(self.comm, hidden.shape[0]), hidden.shape[0] * self.rf_shard.rf.w)

    def sampler(self):
        return _SamplerView(self)

    def check_err(self):
        self.comm.check_err()
