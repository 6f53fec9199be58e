"""Execution engine of the PianoBART hot path on MI355X.

Owns (a) one flat f32 parameter buffer + one flat f32 gradient buffer (+ Adam moments and a bf16
weight shadow) whose slices are what the nn.Parameters of pianobart_amd.model alias, (b) the saved
activations of one forward, and (c) the explicit forward / backward schedules, written as sequences
of C-ABI kernel launches on torch's current HIP stream (no autograd graph inside, no tracing).

Reference semantics followed (file:line into /root/reference; tf: = transformers modeling_bart.py):
  embed:   PianoBart.py:60-71 + tf:520-525/648-654   -> P = 16 E_i W_i^T once per step, gather-sum + pos + LN
  layers:  tf:280-308 (encoder, post-LN), tf:343-390 (decoder: self, cross, ffn)
  heads:   model.py:119-126  -> one (d x 1280) GEMM
  loss:    pretrain.py:112-118,163-189 ; step: pretrain.py:192-196 (clip 3.0, HF AdamW)
"""
import math
import os
import time

import numpy as np
import torch

from . import ops, rowpack

from ._lib import LIB, PB_BF16, PB_F32, PB_F32X3, PBError

_WGRAD_STREAM = int(os.environ.get('PB_WGRAD_STREAM', '7'))            # second HIP stream, bits: 1 = weight-gradient GEMMs, 2 = cross-attention K/V projections, 4 = backward GEMMs as ordinary grids (0: everything on one stream, for A/B)
_WG_DXD_256 = int(os.environ.get('PB_WG_DXD_256', '1'))                         # the d x d weight gradients on the 256x256 ping-pong kernel (9 tiles x split-K 21) instead of 128x128 tiles (36 x 14); same-box A/B round 5: 58.01 / 58.01 -> 57.83 / 57.62 ms (profiles/r05_dxd_wgrad_ab.txt); 0 = the round-2 choice
_WG_TARGET = 192 if _WGRAD_STREAM & 1 else 256            # split-K work items a weight-gradient GEMM aims for (256x256 tiles)
_NO_DEFER = False                  # settled (round 2): True reduces every bias / LayerNorm gradient right behind its producer
_DECODE_SPLIT = True               # settled (round 2): False = single-query attention with one workgroup per head
_ROWDOT = int(os.environ.get('PB_ROWDOT', '1'))                                  # 1 = delta of the one-pass attention backward from the out-projection dgrad's epilogue (0: a separate pass)
_X3_ONEHOT = int(os.environ.get('PB_X3_ONEHOT', '1'))                        # bf16x3: embedding-table gradient as two one-hot GEMMs over dz's (hi, lo) planes (0: the exact path's f32 atomics, for A/B)
_DP_RESERVE_CUS = int(os.environ.get('PB_DP_RESERVE_CUS', '0'))                 # data parallel: 0 = backward GEMMs as ordinary grids (default: +0.45 ms at world 1, profiles/r06_dp_mode_ab.txt); n > 0 = persistent grids that leave n CUs to RCCL's kernels (+1.0 / +1.4 ms for 8 / 16)
_SIDE_PRIORITY = int(os.environ.get('PB_SIDE_PRIORITY', '0'))                      # HIP priority of the second stream (1 = low, -1 = high; developer A/B)
_X3_FLASH = int(os.environ.get('PB_X3_FLASH', '1'))                            # bf16x3: 1 = fused split-bf16 attention (pb_flash_*_x3), 0 = the unfused QK^T / softmax / PV chain of the exact-f32 path
_DECODE_SPEC = int(os.environ.get('PB_DECODE_SPEC', '1'))                       # 1 = device-side sampling ahead of the host where the caller names the sampler (Engine._generate_device_sampled), 0 = one host round trip per token
_DECODE_GRAPH = int(os.environ.get('PB_DECODE_GRAPH', '1'))                     # 1 = one hipGraph replay per token (6 launches per layer), 0 = the same launches issued directly, -1 = the round-2 per-launch loop (the persistent-kernel forms of round 4, measured slower, left the library in round 5: profiles/r04_decode_persistent.txt)
_NO_FUSED_BIAS = False             # settled (round 2): True takes the bias gradients out of the GEMM / attention epilogues

LN_EPS = 1e-5


# copies of the backward scratch buffers that the second stream's weight-gradient GEMMs read (gB, du, dq, dkv, dqkv): with 2 the main
# stream can write the next sublayer's cotangent while the weight gradient of the previous one is still reading its own (3: slower)
_RING = 2
# events that order the two streams: HIP events without the system-scope fence (hipEventDisableSystemFence: 59.5 -> 59.1 ms/step
# against plain hipEventDisableTiming events; hipEventReleaseToDevice: no gain)
_EVENT_MODE = 1
_SIDE_TAIL = 1   # settled (round 2). End of backward: decoder half of dP, the deferred reductions and one f32 GEMM on the second stream
_DGRAD_NT = 1    # settled (round 2): backward dX = dY W from transposed weight copies (NT GEMM) instead of the NN form
_FWD_GEMM_FLAGS = int(os.environ.get('PB_FWD_GEMM_FLAGS', '32768'))      # PB_GEMM_TAIL_SPLIT for the forward projections (0: off)
# dead-row compaction of the fused pre-train step (Engine._pack_batch): PB_PACK_ROWS=0 keeps every step dense
_PACK_ROWS = int(os.environ.get('PB_PACK_ROWS', '1'))
# head_dim-64 attention backward in ONE pass (csrc/pb_flash1.hip) instead of the dQ + dK/dV kernel pair: 0 = never, 1 = the padded
# (dense) non-causal calls only (11 - 14 % faster per call), 2 = packed non-causal calls too (default: 3 % faster per call alone, 0.9 ms
# of the 59 ms step in a same-box A/B), 3 = causal calls too (slower: a key block's waves idle through the masked half of the diagonal)
_ATTN_BWD1 = int(os.environ.get('PB_ATTN_BWD1', '2'))
_KV_CUT = 2      # the stacked cross-attention K / V projection is issued as layers 0 .. _KV_CUT - 1 | the rest (forward_hidden)


class _HipEvent:
    """One HIP event of the engine's pool (pb_event_*): record() on the current stream, wait_on(stream)."""
    __slots__ = ('h',)

    def __init__(self, mode):
        import ctypes
        self.h = ctypes.c_void_p()
        LIB.call('pb_event_create', ctypes.byref(self.h), mode)

    def record(self):
        import ctypes
        LIB.call('pb_event_record', self.h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        return self

    def wait_on(self, stream):
        import ctypes
        LIB.call('pb_stream_wait_event', ctypes.c_void_p(stream.cuda_stream), self.h)


def _r4(n):
    return (n + 3) // 4 * 4


class _Slot:
    __slots__ = ('off', 'shape', 'numel')

    def __init__(self, off, shape):
        self.off, self.shape = off, tuple(shape)
        self.numel = int(np.prod(shape))


def _new_stream(device):
    """A HIP stream for the engine's second stream. PB_SIDE_PRIORITY=1 (developer A/B) makes it a LOW-priority one: torch clamps priorities to
    [-1, 0], HIP has -1 / 0 / 1, so the stream is created through the runtime and wrapped."""
    if _SIDE_PRIORITY == 0:
        return torch.cuda.Stream(device=device)
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    st = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipStreamCreateWithPriority(ctypes.byref(st), 1, _SIDE_PRIORITY)          # 1 = hipStreamNonBlocking
    if rc != 0 or not st.value:
        return torch.cuda.Stream(device=device)
    return torch.cuda.ExternalStream(st.value, device=device)


class Engine:
    def __init__(self, pianobart, mask_lm, precision='bf16'):
        if precision not in ('bf16', 'fp32', 'bf16x3'):
            raise PBError('precision must be "bf16", "fp32" or "bf16x3"')
        self.pb, self.mlm = pianobart, mask_lm
        for mod in (pianobart, mask_lm):
            if mod is not None and hasattr(mod, 'register_state_dict_pre_hook'):         # a checkpoint must not read parameters a pipelined update is still writing
                mod.register_state_dict_pre_hook(lambda *a, **k: self.finish_updates())
        self.precision = precision
        self.code = PB_BF16 if precision == 'bf16' else PB_F32
        self.xdt = torch.bfloat16 if precision == 'bf16' else torch.float32
        # "bf16x3" (round 6): the exact-f32 instantiation -- f32 storage, f32 LayerNorm / softmax / loss / dropout kernels, unfused attention --
        # with every GEMM handed to pb_gemm as PB_F32X3: f32 operands cut into bf16 (hi, lo) pairs, a_hi b_hi + a_hi b_lo + a_lo b_hi on
        # the bf16 matrix cores with f32 accumulation (pb_gemm_x3.hip). ~2^-16 per product instead of bf16's 2^-8: the parity-grade
        # instantiation (logits <= 1e-3 of the CPU reference, north_star) that does not run at the f32-input MFMA rate.
        self.x3 = precision == 'bf16x3'
        self.gcode = PB_F32X3 if self.x3 else self.code
        cfg = pianobart.bartConfig
        self.cfg = cfg
        self.d = cfg.d_model
        self.H = cfg.encoder_attention_heads
        self.hd = self.d // self.H
        self.NE, self.ND = cfg.encoder_layers, cfg.decoder_layers
        self.fe, self.fd = cfg.encoder_ffn_dim, cfg.decoder_ffn_dim
        self.Smax = cfg.max_position_embeddings
        self.p_drop = float(getattr(cfg, 'dropout', 0.1))
        self.device = None
        self._layout()
        self._ws_cache = {}
        self._saved = None
        self._fwd_token = 0
        # dropout stream: one LCG walk per process, keyed by the rank (data-parallel replicas must not share dropout masks)
        self._seed = (0x5EED1234 + 0x9E3779B97F4A7C15 * int(os.environ.get('RANK', 0))) & 0xFFFFFFFFFFFFFFFF
        self.step_count = 0
        self.opt_m = self.opt_v = None
        self._versions = None
        self.use_flash = (self.code == PB_BF16 and self.hd in (32, 64, 96, 128)) or (self.x3 and _X3_FLASH and self.hd in (32, 64, 128))
        self._slabs = None
        self._pack_state = None
        self._pack_prefetch, self._pack_pf_state = [], None     # rowpack.prefetch_counts: requests in flight (oldest first)
        self._ev_pool = None
        self._slabs_oh = [None, None]
        self.pipeline_updates = False       # optimizer_step may leave the parameter update running on the second stream (see there)
        self._upd, self._upd_waited, self._upd_groups = None, 0, None
        self._tables_ready = False          # forward_hidden's table work was already issued by _pack_batch for this step
        self.last_rows = self.last_pairs = None
        self._side, self._side_last, self._readers = None, None, {}
        self._kmax = {}
        self.grad_hook = None          # callable(lo, hi): flat gradient range is final (data-parallel bucketing)

    # ------------------------------------------------------------------ flat parameter layout
    def _layout(self):
        d, fe, fd = self.d, self.fe, self.fd
        slots, cur = {}, [0]

        def add(name, *shape):
            slots[name] = _Slot(cur[0], shape)
            cur[0] += _r4(int(np.prod(shape)))

        # region A: matrices whose gradients are OVERWRITTEN by one wgrad GEMM each step
        add('emb', ops.TAB_TOTAL, 256)                 # 8 slots of 264 rows (pad rows stay zero: zero grad, zero decay)
        add('lin.w', d, 2048)
        for l in range(self.NE):
            p = 'enc.%d.' % l
            add(p + 'wqkv', 3 * d, d); add(p + 'wo', d, d); add(p + 'w1', fe, d); add(p + 'w2', d, fe)
        for l in range(self.ND):
            p = 'dec.%d.' % l
            add(p + 'wqkv', 3 * d, d); add(p + 'wo', d, d); add(p + 'wq_c', d, d)
            add(p + 'wo_c', d, d); add(p + 'w1', fd, d); add(p + 'w2', d, fd)
        # the cross-attention K / V projections of ALL decoder layers form one (ND * 2d, d) matrix: every layer projects the same
        # encoder output, so forward, input-gradient and weight-gradient are one long GEMM each instead of ND short ones
        # (tf:modeling_bart.py:227-228, the key_value_states branch). 'dec.<l>.wkv_c' / 'dec.<l>.bkv_c' stay as row-slice aliases.
        if self.ND:
            add('dec.wkv_all', self.ND * 2 * d, d)
        if self.mlm is not None:
            add('head.w', ops.VOCAB, d)
        self.n_matrix = cur[0]
        # region B: vectors / tables whose gradients are ACCUMULATED (zeroed at the start of each backward)
        add('lin.b', d)
        for side, n in (('enc', self.NE), ('dec', self.ND)):
            add(side + '.pos', self.Smax + 2, d); add(side + '.lne.w', d); add(side + '.lne.b', d)
            for l in range(n):
                p = '%s.%d.' % (side, l)
                add(p + 'bqkv', 3 * d); add(p + 'bo', d); add(p + 'ln1.w', d); add(p + 'ln1.b', d)
                if side == 'dec':
                    add(p + 'bq_c', d); add(p + 'bo_c', d); add(p + 'lnc.w', d); add(p + 'lnc.b', d)
                add(p + 'b1', fe if side == 'enc' else fd); add(p + 'b2', d); add(p + 'ln2.w', d); add(p + 'ln2.b', d)
        if self.ND:
            add('dec.bkv_all', self.ND * 2 * d)
        if self.mlm is not None:
            add('head.b', ops.VOCAB)
        self.n_total = cur[0]
        self.slots = slots

    def _param_map(self):
        """[(nn.Parameter, slot name, row offset in slot)] for every live parameter, in a fixed order."""
        pb, d = self.pb, self.d
        out = []
        off = 0
        for i in range(8):
            out.append((pb.word_emb[i].lut.weight, 'emb', ops.TAB_OFF[i]))
        out.append((pb.encoder_linear.weight, 'lin.w', 0)); out.append((pb.encoder_linear.bias, 'lin.b', 0))
        for side, stack, n in (('enc', pb.bart.encoder, self.NE), ('dec', pb.bart.decoder, self.ND)):
            out.append((stack.embed_positions.weight, side + '.pos', 0))
            out.append((stack.layernorm_embedding.weight, side + '.lne.w', 0)); out.append((stack.layernorm_embedding.bias, side + '.lne.b', 0))
            for l in range(n):
                L, p = stack.layers[l], '%s.%d.' % (side, l)
                sa = L.self_attn
                out += [(sa.q_proj.weight, p + 'wqkv', 0), (sa.k_proj.weight, p + 'wqkv', d), (sa.v_proj.weight, p + 'wqkv', 2 * d),
                        (sa.q_proj.bias, p + 'bqkv', 0), (sa.k_proj.bias, p + 'bqkv', d), (sa.v_proj.bias, p + 'bqkv', 2 * d),
                        (sa.out_proj.weight, p + 'wo', 0), (sa.out_proj.bias, p + 'bo', 0),
                        (L.self_attn_layer_norm.weight, p + 'ln1.w', 0), (L.self_attn_layer_norm.bias, p + 'ln1.b', 0)]
                if side == 'dec':
                    ca = L.encoder_attn
                    out += [(ca.q_proj.weight, p + 'wq_c', 0), (ca.q_proj.bias, p + 'bq_c', 0),
                            (ca.k_proj.weight, 'dec.wkv_all', l * 2 * d), (ca.v_proj.weight, 'dec.wkv_all', l * 2 * d + d),
                            (ca.k_proj.bias, 'dec.bkv_all', l * 2 * d), (ca.v_proj.bias, 'dec.bkv_all', l * 2 * d + d),
                            (ca.out_proj.weight, p + 'wo_c', 0), (ca.out_proj.bias, p + 'bo_c', 0),
                            (L.encoder_attn_layer_norm.weight, p + 'lnc.w', 0), (L.encoder_attn_layer_norm.bias, p + 'lnc.b', 0)]
                out += [(L.fc1.weight, p + 'w1', 0), (L.fc1.bias, p + 'b1', 0), (L.fc2.weight, p + 'w2', 0), (L.fc2.bias, p + 'b2', 0),
                        (L.final_layer_norm.weight, p + 'ln2.w', 0), (L.final_layer_norm.bias, p + 'ln2.b', 0)]
        if self.mlm is not None:
            off = 0
            for i in range(8):
                out.append((self.mlm.proj[i].weight, 'head.w', off)); out.append((self.mlm.proj[i].bias, 'head.b', off))
                off += ops.SEG_SIZES[i]
        return out

    def _elem_off(self, slot, row):
        s = self.slots[slot]
        inner = int(np.prod(s.shape[1:])) if len(s.shape) > 1 else 1
        return s.off + row * inner

    # ------------------------------------------------------------------ binding params to the flat buffers
    def bind(self, device):
        """(Re)build the flat buffers on `device` and make every nn.Parameter alias its slice."""
        pm = self._param_map()
        if self.device is not None and all(p.data_ptr() == self.P32.data_ptr() + 4 * self._elem_off(s, r) for p, s, r in pm):
            return
        pdev = pm[0][0].device
        if device.type != 'cuda' or pdev.type != 'cuda':
            raise PBError('pianobart_amd runs on the HIP device only (model on %s, inputs on %s). Move the model with '
                          '.to("cuda"); there is no CPU execution path.' % (pdev, device))
        self.device = device
        P32 = torch.zeros(self.n_total, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, s, r in pm:
                o = self._elem_off(s, r)
                P32[o:o + p.numel()].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
            for p, s, r in pm:
                o = self._elem_off(s, r)
                p.data = P32[o:o + p.numel()].view(p.shape)
        self.P32 = P32
        self.G32 = torch.zeros(self.n_total, dtype=torch.float32, device=device)
        self.G32_alt = None
        self.Pbf = torch.empty(self.n_total, dtype=torch.bfloat16, device=device) if self.code == PB_BF16 else None
        self.params = [p for p, _, _ in pm]
        self.param_slots = [(s, r) for _, s, r in pm]
        self.grad_views = [self.G32[self._elem_off(s, r):self._elem_off(s, r) + p.numel()].view(p.shape) for p, s, r in pm]
        self.w, self.wf, self.g = {}, {}, {}
        for name, s in self.slots.items():
            self.wf[name] = P32[s.off:s.off + s.numel].view(s.shape)
            self.g[name] = self.G32[s.off:s.off + s.numel].view(s.shape)
            self.w[name] = (self.Pbf[s.off:s.off + s.numel].view(s.shape) if self.code == PB_BF16 else self.wf[name])
        for dct in (self.w, self.wf, self.g):
            self._alias_cross_kv(dct)
        # transposed bf16 copies of the layer matrices for the backward's dX = dY W (read K-contiguous, the faster form of the GEMM
        # kernel): same offsets in a second flat buffer, rewritten by one batched launch whenever the shadow changed
        self.wT, self._wT_table, self._wT_tiles = {}, None, 0
        self._shadow_gen, self._wT_gen, self._wT_done = 0, -1, None
        if self.code == PB_BF16 and _DGRAD_NT:
            self.PbfT = torch.empty_like(self.Pbf)
            rows, tiles = [], 0
            for name, s in self.slots.items():
                if len(s.shape) == 2 and s.off < self.n_matrix and name not in ('emb', 'lin.w') and s.shape[0] % 8 == 0 and s.shape[1] % 8 == 0 \
                        and s.off % 8 == 0:
                    R, C = s.shape
                    rows.append([s.off, R, C, tiles])
                    tiles += -(-R // 64) * -(-C // 64)
                    self.wT[name] = self.PbfT[s.off:s.off + s.numel].view(C, R)
            if rows:
                self._wT_table = torch.tensor(rows, dtype=torch.int32, device=device)
                self._wT_tiles = tiles
        self.Gcur = self.G32
        self.opt_m = self.opt_v = None
        self._versions = None
        self._ws_cache = {}
        self.ptab = torch.zeros(ops.TAB_TOTAL, self.d, dtype=torch.float32, device=device)
        self.dptab = torch.zeros(ops.TAB_TOTAL, self.d, dtype=torch.float32, device=device)
        npart = max(int(LIB.query('pb_ln_partials_floats', self.d)), int(LIB.query('pb_colsum_partials_floats', max(3 * self.d, self.fe, self.fd, ops.VOCAB))),
                    int(LIB.query('pb_ce_partials_floats')), int(LIB.query('pb_norm_partials_floats')))
        self.partials = torch.empty(npart, dtype=torch.float32, device=device)
        self.scal = torch.zeros(64, dtype=torch.float32, device=device)    # [0:24] ce sums, [24:32] counts, [32:40] coef, [40] sq, [41] clip
        w = [len(self.pb.e2w[k]) for k in self.pb.e2w]                       # dict order (pretrain.py:185-189)
        self.loss_w = torch.tensor(w, dtype=torch.float32, device=device)
        self.sos16 = torch.tensor(self.pb.sos_word_np, dtype=torch.int16, device=device)

    def _alias_cross_kv(self, dct):
        """'dec.<l>.wkv_c' (2d, d) / 'dec.<l>.bkv_c' (2d,) = layer l's rows of the stacked cross-attention K / V projection."""
        if self.ND:
            d2 = 2 * self.d
            for l in range(self.ND):
                dct['dec.%d.wkv_c' % l] = dct['dec.wkv_all'][l * d2:(l + 1) * d2]
                dct['dec.%d.bkv_c' % l] = dct['dec.bkv_all'][l * d2:(l + 1) * d2]

    def refresh_shadow(self, force=False):
        """bf16 weight shadow follows the f32 masters (after load_state_dict / an external optimizer)."""
        if self.code != PB_BF16:
            return
        ver = sum(p._version for p in self.params)
        if force or ver != self._versions:
            self._await_updates(2)
            ops.cast_f32_to_bf16(self.P32, self.Pbf)
            self._versions = ver
            self._shadow_gen += 1

    def _refresh_wT(self, on_side=False):
        """Rewrite the transposed weight copies from the current shadow. on_side: on the second stream (after the optimizer, so that
        it runs beside the next forward); the backward waits for it in _ensure_wT."""
        if self._wT_table is None:
            return
        gen = self._shadow_gen
        if on_side and self._side and _WGRAD_STREAM:
            self._event().wait_on(self._side)
            with torch.cuda.stream(self._side):
                ops.transpose_batch_bf16(self.Pbf, self.PbfT, self._wT_table, self._wT_tiles)
                self._wT_done = self._event()
        else:
            ops.transpose_batch_bf16(self.Pbf, self.PbfT, self._wT_table, self._wT_tiles)
            self._wT_done = None
        self._wT_gen = gen

    def _ensure_wT(self):
        if self._wT_table is None:
            return
        if self._wT_gen != self._shadow_gen:
            self._refresh_wT()
        if self._wT_done is not None:
            self._wT_done.wait_on(torch.cuda.current_stream())
            self._wT_done = None

    # ------------------------------------------------------------------ workspace
    def _ws(self, B, S):
        key = (B, S)
        ws = self._ws_cache.get(key)
        if ws is not None:
            return ws
        dev, X, d, H = self.device, self.xdt, self.d, self.H
        T = B * S
        e = lambda *shape, dt=X: torch.empty(*shape, dtype=dt, device=dev)
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)

        def attn_ws():
            return dict(lse=f(B, H, S)) if self.use_flash else dict(P=e(B, H, S, S))

        def layer(dec):
            ff = self.fd if dec else self.fe
            L = dict(qkv=e(T, 3 * d), ctx=e(T, d), a1=e(T, d), y1=e(T, d), m1=f(T), r1=f(T), u=e(T, ff), g=e(T, ff), a2=e(T, d),
                     y2=e(T, d), m2=f(T), r2=f(T), attn=attn_ws())
            if dec:
                L.update(qc=e(T, d), ctxc=e(T, d), ac=e(T, d), yc=e(T, d), mc=f(T), rc=f(T), attnc=attn_ws())
            return L

        ws = dict(B=B, S=S, T=T,
                  x_enc=e(T, d), me=f(T), re=f(T), x_dec=e(T, d), md=f(T), rd=f(T),
                  kvc_all=e(T, max(1, self.ND) * 2 * d), dkv_all=e(T, max(1, self.ND) * 2 * d),      # cross-attention K | V of every decoder layer side by side, and their gradients
                  enc=[layer(False) for _ in range(self.NE)], dec=[layer(True) for _ in range(self.ND)],
                  logits=f(T, ops.VOCAB) if self.mlm is not None else None,
                  scores=None if self.use_flash else f(B, H, S, S), dS=None if self.use_flash else e(B, H, S, S), delta=f(B, H, S),
                  gy=[e(T, d), e(T, d)], gA=e(T, d), gB=e(T, d), gC=e(T, d), dqkv=e(T, 3 * d), dq=e(T, d),
                  du=e(T, max(self.fe, self.fd)), genc=e(T, d), dlogits=e(T, ops.VOCAB) if self.mlm is not None else None,
                  dz=e(2 * T, d) if (self.code == PB_BF16 or self.x3) else None,
                  onehot=torch.empty(2 * T, ops.TAB_TOTAL, dtype=torch.bfloat16, device=self.device) if (self.code == PB_BF16 or self.x3) else None,
                  dz_planes=torch.empty(2, 2 * T, d, dtype=torch.bfloat16, device=self.device) if self.x3 else None)    # bf16x3: dz cut into (hi, lo) for the one-hot GEMMs
        ws['Te'] = ws['Td'] = T
        self._ws_cache = {key: ws}          # keep one shape resident
        return ws

    def _ws_rows(self, ws, Te, Td):
        """The workspace cut to Te encoder-side and Td decoder-side rows (packed step): row-prefix views of the same storage. The
        shared scratch buffers (gy, gA, dqkv ...) stay whole; backward cuts them per phase."""
        if Te == ws['T'] and Td == ws['T']:
            return ws
        views = ws.setdefault('_views', {})
        v = views.get((Te, Td))
        if v is None:
            if len(views) >= 64:
                views.clear()

            def layer(L, dec):
                n = Td if dec else Te
                return {k: (t if k in ('attn', 'attnc') else t[:n]) for k, t in L.items()}

            v = dict(ws)
            v['_base'] = ws
            v['enc'] = [layer(L, False) for L in ws['enc']]
            v['dec'] = [layer(L, True) for L in ws['dec']]
            for k in ('x_enc', 'me', 're', 'kvc_all', 'dkv_all'):
                v[k] = ws[k][:Te]
            for k in ('x_dec', 'md', 'rd'):
                v[k] = ws[k][:Td]
            for k in ('logits', 'dlogits'):
                v[k] = ws[k][:Td] if ws[k] is not None else None
            v['Te'], v['Td'] = Te, Td
            views[(Te, Td)] = v
        return v

    # ------------------------------------------------------------------ building blocks
    def _linear(self, x, wname, bname, out, M, N, K, **kw):
        # forward GEMMs run alone on their stream (nothing fills a partly filled last round of the persistent grid): allow the tail split
        ops.gemm(x, self.w[wname], out, M=M, N=N, K=K, dtype=self.gcode, bias=self.wf[bname] if bname else None, dbg=_FWD_GEMM_FLAGS, **kw)

    def _attn_fwd(self, q, k, v, out, key_mask, causal, B, Sq, Sk, save, rows=None):
        """q,k,v,out: (tensor, elem offset, row stride). Unfused form: QK^T -> masked softmax -> PV. rows: packed-row descriptors."""
        H, hd = self.H, self.hd
        ws = self._cur_ws
        if rows is not None and self.x3:
            ops.flash_fwd_x3_packed(q, k, v, out, save['lse'], rows, B, H, hd, hd ** -0.5, causal)
            return
        if rows is not None:
            ops.flash_fwd_packed(q, k, v, out, save['lse'], rows, B, H, hd, hd ** -0.5, causal)
            return
        if self.use_flash and self.x3:
            ex = lambda a, n: (a[0], a[1], a[2], n * a[2])
            ops.flash_fwd_x3(ex(q, Sq), ex(k, Sk), ex(v, Sk), ex(out, Sq), save['lse'], key_mask, B, H, Sq, Sk, hd, hd ** -0.5, causal,
                             kmax=self._kmax.get(id(key_mask)) if key_mask is not None else None)
            return
        if self.use_flash:
            ex = lambda a, n: (a[0], a[1], a[2], n * a[2])
            ops.flash_fwd(ex(q, Sq), ex(k, Sk), ex(v, Sk), ex(out, Sq), save['lse'], key_mask, B, H, Sq, Sk, hd, hd ** -0.5, causal,
                          kmax=self._kmax.get(id(key_mask)) if key_mask is not None else None)
            return
        scores, P = ws['scores'], save['P']
        (qt, qo, ql), (kt, ko, kl), (vt, vo, vl), (ot, oo, ol) = q, k, v, out
        ops.gemm(qt, kt, scores, M=Sq, N=Sk, K=hd, dtype=self.gcode, lda=ql, ldb=kl, ldc=Sk, c_f32=True, nb1=B, nb2=H,
                 sA=(Sq * ql, hd), sB=(Sk * kl, hd), sC=(H * Sq * Sk, Sq * Sk), a_off=qo, b_off=ko)
        ops.softmax_fwd(scores, key_mask, P, B, H, Sq, Sk, hd ** -0.5, causal)
        ops.gemm(P, vt, ot, M=Sq, N=hd, K=Sk, dtype=self.gcode, b_kc=False, lda=Sk, ldb=vl, ldc=ol, nb1=B, nb2=H,
                 sA=(H * Sq * Sk, Sq * Sk), sB=(Sk * vl, hd), sC=(Sq * ol, hd), b_off=vo, c_off=oo)

    def _one_pass_bwd(self, causal, rows, B, Sq, Sk, q_rows):
        """Whether the attention backward of this call is the one-pass kernel (head_dim 64; PB_ATTN_BWD1; its -lse / -delta tables of a whole
        sequence live in LDS: beyond 6144 queries the dQ + dK/dV pair takes the call)."""
        if not (self.use_flash and not self.x3 and self.hd == 64 and _ATTN_BWD1 >= (3 if causal else 2 if rows is not None else 1)):
            return False
        return bool(LIB.query('pb_flash_bwd1_supported', rows.Sq_max if rows is not None else Sq, rows.Sk_max if rows is not None else Sk,
                              self.hd, q_rows, self.H))

    def _attn_bwd(self, dout, q, k, v, dq, dk, dv, B, Sq, Sk, save, out=None, key_mask=None, causal=False, dbias=None, rows=None, delta_rows=None):
        """dbias = (gq, gk, gv) bias-gradient vectors: filled here when the attention kernels can do it (returns True), else left to the caller.
        delta_rows: rowsum(dO * O) per head, [H][rows], already made by the GEMM that produced dout (PB_GEMM_ROWDOT); one-pass kernel only."""
        H, hd = self.H, self.hd
        ws = self._cur_ws
        if self.use_flash and self.x3 and rows is not None:
            assert dout[1] == 0 and dout[2] == out[2]
            ops.flash_bwd_x3_packed(q, k, v, out, dout[0], save['lse'], dq, dk, dv, ws['delta'], rows, B, H, hd, hd ** -0.5, causal)
            return False
        if self.use_flash and self.x3:
            ex = lambda a, n: (a[0], a[1], a[2], n * a[2])
            assert dout[1] == 0 and dout[2] == out[2]
            ops.flash_bwd_x3(ex(q, Sq), ex(k, Sk), ex(v, Sk), ex(out, Sq), dout[0], save['lse'], key_mask, ex(dq, Sq), ex(dk, Sk), ex(dv, Sk),
                             ws['delta'], B, H, Sq, Sk, hd, hd ** -0.5, causal, kmax=self._kmax.get(id(key_mask)) if key_mask is not None else None)
            return False
        if self.use_flash:
            ex = lambda a, n: (a[0], a[1], a[2], n * a[2])
            assert dout[1] == 0 and dout[2] == out[2]
            fuse = dbias is not None and hd in (64, 96, 128) and not _NO_FUSED_BIAS
            wsb = None
            if fuse:
                need = int(LIB.query('pb_flash_bias_ws_floats', B, H, Sq, Sk, hd))
                if getattr(self, '_fbws', None) is None or self._fbws.numel() < need:
                    self._fbws = torch.empty(need, dtype=torch.float32, device=self.device)
                wsb = self._fbws
            one_pass = self._one_pass_bwd(causal, rows, B, Sq, Sk, q[0].shape[0] if rows is not None else B * Sq)
            assert one_pass or delta_rows is None
            if rows is not None:
                if one_pass:
                    ops.flash_bwd1_packed(q, k, v, out, dout[0], save['lse'], dq, dk, dv, ws['delta'], rows, B, H, hd, hd ** -0.5, causal, q[0].shape[0],
                                          dbias=dbias if fuse else None, dbias_ws=wsb, delta_rows=delta_rows)
                else:
                    ops.flash_bwd_packed(q, k, v, out, dout[0], save['lse'], dq, dk, dv, ws['delta'], rows, B, H, hd, hd ** -0.5, causal,
                                         dbias=dbias if fuse else None, dbias_ws=wsb)
                return fuse
            (ops.flash_bwd1 if one_pass else ops.flash_bwd)(
                ex(q, Sq), ex(k, Sk), ex(v, Sk), ex(out, Sq), dout[0], save['lse'], key_mask, ex(dq, Sq), ex(dk, Sk), ex(dv, Sk),
                ws['delta'], B, H, Sq, Sk, hd, hd ** -0.5, causal, kmax=self._kmax.get(id(key_mask)) if key_mask is not None else None,
                dbias=dbias if fuse else None, dbias_ws=wsb, **({'delta_rows': delta_rows} if one_pass else {}))
            return fuse
        dP, dS, P = ws['scores'], ws['dS'], save['P']
        (qt, qo, ql), (kt, ko, kl), (vt, vo, vl) = q, k, v
        (dot, doo, dol) = dout
        (dqt, dqo, dql), (dkt, dko, dkl), (dvt, dvo, dvl) = dq, dk, dv
        bs = (H * Sq * Sk, Sq * Sk)
        ops.gemm(dot, vt, dP, M=Sq, N=Sk, K=hd, dtype=self.gcode, lda=dol, ldb=vl, ldc=Sk, c_f32=True, nb1=B, nb2=H,
                 sA=(Sq * dol, hd), sB=(Sk * vl, hd), sC=bs, a_off=doo, b_off=vo)
        ops.softmax_bwd(dP, P, dS, B * H * Sq, Sk, hd ** -0.5)
        ops.gemm(dS, kt, dqt, M=Sq, N=hd, K=Sk, dtype=self.gcode, b_kc=False, lda=Sk, ldb=kl, ldc=dql, nb1=B, nb2=H,
                 sA=bs, sB=(Sk * kl, hd), sC=(Sq * dql, hd), b_off=ko, c_off=dqo)
        ops.gemm(dS, qt, dkt, M=Sk, N=hd, K=Sq, dtype=self.gcode, a_kc=False, b_kc=False, lda=Sk, ldb=ql, ldc=dkl, nb1=B, nb2=H,
                 sA=bs, sB=(Sq * ql, hd), sC=(Sk * dkl, hd), b_off=qo, c_off=dko)
        ops.gemm(P, dot, dvt, M=Sk, N=hd, K=Sq, dtype=self.gcode, a_kc=False, b_kc=False, lda=Sk, ldb=dol, ldc=dvl, nb1=B, nb2=H,
                 sA=bs, sB=(Sq * dol, hd), sC=(Sk * dvl, hd), b_off=doo, c_off=dvo)
        return False

    def _site(self, kind, layer=0, sub=0):
        return {'enc_emb': 0, 'dec_emb': 1}.get(kind, 2 + (layer * 8 + sub) * 2 + (0 if kind == 'enc' else 1))

    def build_ptab(self):
        """P[off_i + v] = 16 * E_i[v] @ W_lin[:, 256 i : 256 i + 256]^T in exact f32 (PianoBart.py:16,67-71)."""
        R = ops.TAB_ROWS
        ops.gemm(self.wf['emb'], self.wf['lin.w'], self.ptab, M=R, N=self.d, K=256, dtype=PB_F32, lda=256, ldb=2048, ldc=self.d, alpha=16.0,
                 c_f32=True, nb1=8, sA=(R * 256, 0), sB=(256, 0), sC=(R * self.d, 0))

    # ------------------------------------------------------------------ forward
    def forward_hidden(self, enc16, dec16, emask, dmask, train, seed, reuse_encoder=False, dec_embeds=None, pack=None):
        """enc16/dec16: (B,S,8) int16 device; masks (B,S) f32 or None. Returns (dec_hidden, enc_hidden) in storage dtype.
        pack (a rowpack.RowPack): enc16 / dec16 hold the packed rows (Te,8) / (Td,8) of the batch and the masks are not read."""
        B, S = (pack.B, pack.S) if pack is not None else enc16.shape[:2]
        if S > self.Smax:
            raise PBError('sequence length %d exceeds max_position_embeddings %d' % (S, self.Smax))
        self._fwd_token += 1          # any forward (generate included) overwrites the activation workspace: older autograd graphs are stale
        d = self.d
        Te, Td = (pack.Te, pack.Td) if pack is not None else (B * S, B * S)     # rows on the encoder / decoder side
        ws = self._ws_rows(self._ws(B, S), Te, Td)
        self._cur_ws = ws
        r_enc, r_dec, r_cross = (pack.enc, pack.dec, pack.cross) if pack is not None else (None, None, None)
        ids_e, ids_d = (pack.src_e, pack.src_d) if pack is not None else (None, None)        # row numbers in the padded batch
        p = self.p_drop if train else 0.0
        self._await_updates(0)
        if not self._tables_ready:
            self.refresh_shadow()
            self.build_ptab()
        self._tables_ready = False
        # per-batch-row key extents (1 + last visible key): the attention kernels skip the masked PAD tail tile-wise
        self._kmax = {}
        if self.use_flash and (self.x3 or self.hd in (64, 96, 128)) and pack is None:
            for msk in (emask, dmask):
                if msk is not None and id(msk) not in self._kmax:
                    km = torch.empty(msk.shape[0], dtype=torch.int32, device=msk.device)
                    ops.key_extent(msk, km)
                    self._kmax[id(msk)] = km
        self._kmax_keep = (emask, dmask)          # keep the mask objects alive while their ids are keys
        wf = self.wf
        x = ws['x_enc']
        if not reuse_encoder:
            ops.embed_ln_fwd(enc16, self.ptab, wf['lin.b'], wf['enc.pos'], wf['enc.lne.w'], wf['enc.lne.b'], x, ws['me'], ws['re'], S,
                             LN_EPS, seed, self._site('enc_emb'), p, padded=True, row_ids=ids_e)
        T = Te
        for l in range(self.NE if not reuse_encoder else 0):
            L, pf = ws['enc'][l], 'enc.%d.' % l
            if l == 2:
                self._await_updates(1)
            self._linear(x, pf + 'wqkv', pf + 'bqkv', L['qkv'], T, 3 * d, d)
            self._attn_fwd((L['qkv'], 0, 3 * d), (L['qkv'], d, 3 * d), (L['qkv'], 2 * d, 3 * d), (L['ctx'], 0, d), emask, False, B, S, S, L['attn'],
                           rows=r_enc)
            self._linear(L['ctx'], pf + 'wo', pf + 'bo', L['a1'], T, d, d)
            ops.add_ln_fwd(x, L['a1'], wf[pf + 'ln1.w'], wf[pf + 'ln1.b'], L['y1'], L['m1'], L['r1'], LN_EPS, seed, self._site('enc', l, 0), p, row_ids=ids_e)
            self._linear(L['y1'], pf + 'w1', pf + 'b1', L['g'], T, self.fe, d, gelu_aux_out=L['u'])
            self._linear(L['g'], pf + 'w2', pf + 'b2', L['a2'], T, d, self.fe)
            ops.add_ln_fwd(L['y1'], L['a2'], wf[pf + 'ln2.w'], wf[pf + 'ln2.b'], L['y2'], L['m2'], L['r2'], LN_EPS, seed, self._site('enc', l, 1), p, row_ids=ids_e)
            x = L['y2']
        enc_out = x if not reuse_encoder else (ws['enc'][-1]['y2'] if self.NE else x)
        if dec16 is None and dec_embeds is None:
            return None, enc_out
        self._await_updates(2)
        y = ws['x_dec']
        T = Td
        if dec_embeds is None:
            ops.embed_ln_fwd(dec16, self.ptab, wf['lin.b'], wf['dec.pos'], wf['dec.lne.w'], wf['dec.lne.b'], y, ws['md'], ws['rd'], S,
                             LN_EPS, seed, self._site('dec_emb'), p, padded=True, row_ids=ids_d)
        else:
            # decoder_inputs_embeds supplied by the caller (velocity task's label embedding, PianoBart.py:65-66): BART adds the
            # learned positions (offset 2), applies layernorm_embedding, then dropout (modeling_bart.py, BartDecoder.forward)
            ws['alt_pos'] = wf['dec.pos'][2:2 + S].to(self.xdt).unsqueeze(0).expand(B, S, d).reshape(T, d).contiguous()
            ws['alt_e'] = dec_embeds
            pre = y if p == 0.0 else ws.setdefault('x_dec_pre', torch.empty_like(y))
            ops.add_ln_fwd(ws['alt_pos'], dec_embeds, wf['dec.lne.w'], wf['dec.lne.b'], pre, ws['md'], ws['rd'], LN_EPS, 0, 0, 0.0)
            if p > 0.0:
                ops.dropout(pre, y, seed, self._site('dec_emb'), p)
        # Every decoder layer's cross-attention K / V projection reads the encoder output only: they are ONE GEMM against the stacked
        # weights, cut in two so that the first layers' share is there when layer 0 asks for it (layers 0 .. 1 | the rest), on the second
        # stream when there is one (it fills the CUs the decoder's N = d GEMMs leave idle)
        kvld = self.ND * 2 * d
        kv_cut = min(_KV_CUT, self.ND)

        def kv_project(l0, l1):
            ops.gemm(enc_out, self.w['dec.wkv_all'], ws['kvc_all'], M=Te, N=(l1 - l0) * 2 * d, K=d, dtype=self.gcode, bias=self.wf['dec.bkv_all'][l0 * 2 * d:],
                     ldc=kvld, b_off=l0 * 2 * d * d, c_off=l0 * 2 * d, dbg=_FWD_GEMM_FLAGS)
        kv_ready = None
        if (_WGRAD_STREAM & 2) and self._side_stream() is not None and self.ND:
            self._event().wait_on(self._side)
            with torch.cuda.stream(self._side):
                kv_project(0, kv_cut)
                kv_ready = [self._event()] * kv_cut
                if self.ND > kv_cut:
                    kv_project(kv_cut, self.ND)
                    kv_ready += [self._event()] * (self.ND - kv_cut)
        elif self.ND:
            kv_project(0, self.ND)
        kv_of = lambda l: ((ws['kvc_all'], l * 2 * d, kvld), (ws['kvc_all'], l * 2 * d + d, kvld))
        sub = pack.sub if pack is not None else None
        sub_layer = None
        for l in range(self.ND):
            L, pf = ws['dec'][l], 'dec.%d.' % l
            self._linear(y, pf + 'wqkv', pf + 'bqkv', L['qkv'], T, 3 * d, d)
            self._attn_fwd((L['qkv'], 0, 3 * d), (L['qkv'], d, 3 * d), (L['qkv'], 2 * d, 3 * d), (L['ctx'], 0, d), dmask, True, B, S, S, L['attn'],
                           rows=r_dec)
            self._linear(L['ctx'], pf + 'wo', pf + 'bo', L['a1'], T, d, d)
            ops.add_ln_fwd(y, L['a1'], wf[pf + 'ln1.w'], wf[pf + 'ln1.b'], L['y1'], L['m1'], L['r1'], LN_EPS, seed, self._site('dec', l, 0), p, row_ids=ids_d)
            # the rest of the layer (cross-attention block, FFN) works on the query side only: in the last layer of a packed step that
            # is the rows with a loss term (pack.sub), gathered out of y1
            Lq, y1, Tq, ids_q, r_x = L, L['y1'], T, ids_d, r_cross
            if sub is not None and l == self.ND - 1:
                Tq, ids_q, r_x = sub.T, sub.src, sub.cross
                Lq = {k: L[k][:Tq] for k in ('qc', 'ctxc', 'ac', 'yc', 'mc', 'rc', 'u', 'g', 'a2', 'y2', 'm2', 'r2')}
                y1 = Lq['y1s'] = self._y1s(ws, L['y1'])[:Tq]
                ops.gather_rows16(L['y1'], sub.idx, y1, Tq, d * y1.element_size())
                Lq['attnc'] = L['attnc']
                sub_layer = Lq
            self._linear(y1, pf + 'wq_c', pf + 'bq_c', Lq['qc'], Tq, d, d)
            if kv_ready is not None and (l == 0 or l == kv_cut):
                kv_ready[l].wait_on(torch.cuda.current_stream())
            kx, vx = kv_of(l)
            self._attn_fwd((Lq['qc'], 0, d), kx, vx, (Lq['ctxc'], 0, d), emask, False, B, S, S, L['attnc'], rows=r_x)
            self._linear(Lq['ctxc'], pf + 'wo_c', pf + 'bo_c', Lq['ac'], Tq, d, d)
            ops.add_ln_fwd(y1, Lq['ac'], wf[pf + 'lnc.w'], wf[pf + 'lnc.b'], Lq['yc'], Lq['mc'], Lq['rc'], LN_EPS, seed, self._site('dec', l, 1), p, row_ids=ids_q)
            self._linear(Lq['yc'], pf + 'w1', pf + 'b1', Lq['g'], Tq, self.fd, d, gelu_aux_out=Lq['u'])
            self._linear(Lq['g'], pf + 'w2', pf + 'b2', Lq['a2'], Tq, d, self.fd)
            ops.add_ln_fwd(Lq['yc'], Lq['a2'], wf[pf + 'ln2.w'], wf[pf + 'ln2.b'], Lq['y2'], Lq['m2'], Lq['r2'], LN_EPS, seed, self._site('dec', l, 2), p, row_ids=ids_q)
            y = Lq['y2']
        self._saved = dict(enc16=enc16, dec16=dec16, emask=emask, dmask=dmask, p=p, seed=seed, enc_out=enc_out, dec_out=y, B=B, S=S,
                           alt=dec_embeds is not None, pack=pack, sub_layer=sub_layer)
        return y, enc_out

    def heads_forward(self, dec_hidden):
        ws = self._cur_ws
        T = dec_hidden.shape[0]
        logits = ws['logits'][:T]
        ops.gemm(dec_hidden, self.w['head.w'], logits, M=T, N=ops.VOCAB, K=self.d, dtype=self.gcode, bias=self.wf['head.b'], c_f32=True)
        return logits

    # ------------------------------------------------------------------ backward
    def _wgrad(self, dy, x, gname, M, N, T, ldy=None, ldx=None, dy_off=0, x_off=0, g_off=0):
        """G[gname] (M,N) = dy(T,M)^T @ x(T,N)  (TN GEMM into the f32 gradient buffer)."""
        nsplit, slabs, big = 1, None, False
        if (self.code == PB_BF16 or self.x3) and T % 64 == 0:
            big = M >= 256 and N >= 256 and (M * N > 768 * 768 or _WG_DXD_256)     # 256x256 tiles, one block per CU (768x768: 128x128 tiles measured 712 vs 636 TF)
            tl = 256 if big else 128
            tiles = ((M + tl - 1) // tl) * ((N + tl - 1) // tl)
            nsplit = max(1, min(32, T // 64, round((_WG_TARGET if big else 512) / tiles)))
            if nsplit > 1:
                need = nsplit * M * N
                if self._slabs is None or self._slabs.numel() < need:
                    self._join_side()
                    self._slabs = torch.empty(need, dtype=torch.float32, device=self.device)
                slabs = self._slabs
        launch = lambda dbg: ops.gemm(dy, x, self.g[gname], M=M, N=N, K=T, dtype=self.gcode, a_kc=False, b_kc=False, lda=ldy or M, ldb=ldx or N, ldc=N,
                                      c_f32=True, a_off=dy_off, b_off=x_off, c_off=g_off, splitk=nsplit, slabs=slabs, tile256=big, dbg=dbg)
        if not (_WGRAD_STREAM & 1) or self._side_stream() is None:
            return launch(self._bwd_dbg())
        # the weight gradient is off the critical path of backward: run it on a second stream, where its workgroups fill the CUs the
        # main stream's kernels leave idle (the half-empty last round of the N = d GEMMs). Writers of `dy` wait in _before_write.
        self._event().wait_on(self._side)
        with torch.cuda.stream(self._side):
            launch(self._bwd_dbg())
            done = self._event()
        self._readers[dy.untyped_storage().data_ptr()] = done
        self._side_last = done

    def _side_stream(self):
        """A second HIP stream that really runs beside the current one, or None. HIP multiplexes streams onto a few hardware queues
        (4 by default) and two streams that share one execute strictly one after the other, so candidates are probed once: a short
        kernel on the candidate must finish while a ~2 ms train of kernels on the current stream is still running."""
        if self._side is None:
            self._side = False
            main = torch.cuda.current_stream(self.device)
            try:
                buf = torch.empty(256 << 20, dtype=torch.float32, device=self.device)  # 1 GiB: each fill runs ~0.2 ms, far longer than it takes to enqueue
            except RuntimeError:                                                       # no room for the probe: stay on one stream
                return None
            tiny = torch.empty(64, dtype=torch.float32, device=self.device)
            self._side_pool = []
            for _ in range(12):
                cand = _new_stream(self.device)
                self._side_pool.append(cand)            # keep the rejected ones referenced so the pool hands out a different stream next
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                torch.cuda.synchronize(self.device)
                e0.record(main)
                for _ in range(8):
                    ops.fill_f32(buf, 0.0)
                e1.record(main)
                cand.wait_event(e0)
                with torch.cuda.stream(cand):
                    ops.fill_f32(tiny, 0.0)
                    e2.record(cand)
                torch.cuda.synchronize(self.device)
                self._side_probe = (len(self._side_pool), e0.elapsed_time(e2), e0.elapsed_time(e1))
                if self._side_probe[1] < 0.85 * self._side_probe[2]:      # a shared hardware queue gives 1.0 (the candidate's kernel runs after the train)
                    self._side = cand
                    break
            if os.environ.get('PB_DEBUG'):
                print('side stream probe (candidates tried, ms to the candidate kernel, ms of the main train):', self._side_probe, flush=True)
        return self._side or None

    def _before_write(self, *tensors):
        """The main stream is about to overwrite these buffers: wait for the side-stream weight-gradient GEMMs still reading them."""
        if self._readers:
            for t in tensors:
                if t is not None:
                    ev = self._readers.pop(t.untyped_storage().data_ptr(), None)
                    if ev is not None:
                        ev.wait_on(torch.cuda.current_stream())

    def _join_side(self):
        if self._side_last is not None:
            self._side_last.wait_on(torch.cuda.current_stream())
            self._side_last = None
        self._readers.clear()

    def _event(self):
        """A recorded event on the current stream (from a round-robin pool far longer than the events a step has in flight)."""
        pool = self._ev_pool
        if pool is None:
            pool = self._ev_pool = [[_HipEvent(_EVENT_MODE) for _ in range(2048)], 0]
        pool[1] = (pool[1] + 1) % len(pool[0])
        return pool[0][pool[1]].record()

    def _y1s(self, ws, like):
        base = ws.get('_base', ws)
        if 'y1s' not in base:
            base['y1s'] = torch.empty(base['T'], self.d, dtype=like.dtype, device=like.device)
        return base['y1s']

    def _ring(self, name):
        """The next copy of backward scratch buffer `name` (round robin over _RING copies) when weight gradients run on the second
        stream; the one buffer otherwise."""
        ws = self._cur_ws
        base = ws.get('_base', ws)
        if _RING < 2 or not (_WGRAD_STREAM & 1) or not self._side:
            return base[name]
        ring = base.setdefault('_ring', {})
        ent = ring.get(name)
        if ent is None:
            ent = ring[name] = [[base[name]] + [torch.empty_like(base[name]) for _ in range(_RING - 1)], 0]
        ent[1] = (ent[1] + 1) % len(ent[0])
        return ent[0][ent[1]]

    def _dgrad(self, dy, wname, out, T, N, K, accum, ldy=None, **kw):
        """out(T,N) (+)= dy(T,K) @ W(K,N)   (NN GEMM, W stored [K][N])."""
        self._before_write(out)
        wT = self.wT.get(wname)
        if wT is not None:                                        # W^T (N,K): both operands K-contiguous
            ops.gemm(dy, wT, out, M=T, N=N, K=K, dtype=self.gcode, lda=ldy or K, ldb=K, ldc=N, accum=accum, dbg=self._bwd_dbg(), **kw)
            return
        ops.gemm(dy, self.w[wname], out, M=T, N=N, K=K, dtype=self.gcode, b_kc=False, lda=ldy or K, ldb=N, ldc=N, accum=accum, dbg=self._bwd_dbg(), **kw)

    def _cs_ws(self, M, N):
        need = int(LIB.query('pb_gemm_colsum_ws_floats', M, N))
        if getattr(self, '_csbuf', None) is None or self._csbuf.numel() < need:
            self._csbuf = torch.empty(need, dtype=torch.float32, device=self.device)
        return self._csbuf

    def _bwd_dbg(self):
        """Backward GEMMs run beside in-flight gradient all-reduces when a grad_hook is set (data parallel). A persistent
        one-workgroup-per-CU grid whose CUs are partly held by RCCL's kernels would run its stragglers as a second full round, so either
        they are launched as ordinary grids (bit 12, the default), which merely lose those CUs' share, or -- PB_DP_RESERVE_CUS = n > 0 -- the
        persistent grids leave RCCL n CUs up front (GradReducer calls pb_gemm_reserve_cus). Measured at world size 1 (profiles/r06_dp_mode_ab.txt):
        ordinary grids + f32 exchange cost the step +0.45 ms, 8 / 16 reserved CUs +1.0 / +1.4 ms."""
        if self.grad_hook is not None and _DP_RESERVE_CUS > 0:
            return 262144                                            # PB_GEMM_LEAVE_CUS
        return 4096 if (self.grad_hook is not None or ((_WGRAD_STREAM & 4) and self._side)) else 0

    def _ffn_ln_bwd(self, L, pf, ff, gy, y_in, seed, site, p, T, row_ids=None):
        """Backward of y2 = LN2(y_in + drop(fc2(gelu(fc1(y_in))))). gy: grad wrt y2. Returns grad wrt y_in in ws['gA']. T: rows."""
        ws, g, d = self._cur_ws, self.g, self.d
        gA, gB = ws['gA'][:T], self._ring('gB')[:T]
        da = gB if p > 0 else None
        self._before_write(gA, da)
        ops.add_ln_bwd(gy, y_in, L['a2'], self.wf[pf + 'ln2.w'], L['m2'], L['r2'], gA, da, g[pf + 'ln2.w'], g[pf + 'ln2.b'], g[pf + 'b2'],
                       self.partials, False, seed, site, p, row_ids=row_ids)
        gb = gB if p > 0 else gA
        self._wgrad(gb, L['g'], pf + 'w2', d, ff, T)
        du = self._ring('du')
        du = du[:T, :ff] if du.shape[1] == ff else du.view(-1)[:T * ff].view(T, ff)
        # dU = (dG W2) * gelu'(U), and db1 = column sums of dU straight from the same epilogue registers
        if _NO_FUSED_BIAS:
            self._dgrad(gb, pf + 'w2', du, T, ff, d, False, gelu_grad_aux_in=L['u'], ldaux=ff)
            ops.colsum(du, g[pf + 'b1'], self.partials, T, ff)
        else:
            self._dgrad(gb, pf + 'w2', du, T, ff, d, False, gelu_grad_aux_in=L['u'], ldaux=ff, colsum_out=g[pf + 'b1'], colsum_ws=self._cs_ws(T, ff))
        self._wgrad(du, y_in, pf + 'w1', ff, d, T)
        self._dgrad(du, pf + 'w1', gA, T, d, ff, True)
        return gA

    def _ready(self, first, last=None):
        if self.grad_hook is not None and self.Gcur is self.G32:
            a, b = self.slots[first], self.slots[last or first]
            if self._side_last is None:
                return self.grad_hook(a.off, b.off + b.numel)
            # the range was produced by both streams: let the second stream catch up with this one and issue the exchange from it
            # (the collective's own stream orders itself after the stream that is current at the call), so this one never waits
            self._event().wait_on(self._side)
            with torch.cuda.stream(self._side):
                self.grad_hook(a.off, b.off + b.numel)

    def _attn_block_bwd(self, L, pf, names, gy, x_in, q, k, v, dq, dk, dv, ctx, a, attn_save, mean, rstd, lnw, lnb_g, lnw_g, gout, seed, site, p, B, Sq, Sk, key_mask, causal, dbias=None,
                        T=None, rows=None, row_ids=None):
        """Backward of y = LN(x_in + drop(out_proj(attn(q,k,v)))) up to dq/dk/dv. gy: grad wrt y; gout receives grad wrt x_in
        (residual path only; the projection paths are added by the caller)."""
        ws, g, d = self._cur_ws, self.g, self.d
        gB, gC = self._ring('gB')[:T], ws['gC'][:T]         # T: rows on the query side
        wo, bo = names
        da = gB if p > 0 else None
        self._before_write(gout, da, dq[0], dk[0], dv[0])
        ops.add_ln_bwd(gy, x_in, a, lnw, mean, rstd, gout, da, lnw_g, lnb_g, g[bo], self.partials, False, seed, site, p, row_ids=row_ids)
        gb = gB if p > 0 else gout
        self._wgrad(gb, ctx, wo, d, d, T)
        # the one-pass attention backward needs delta = rowsum(dO * O) per head: dO is made right here, so the row sums come out of this GEMM's
        # epilogue (PB_GEMM_ROWDOT: the O tile rides in as the epilogue's operand) instead of a pass over dO and O in front of the attention kernel
        delta_rows = None
        if _ROWDOT and self.code == PB_BF16 and T % 256 == 0 and d % 256 == 0 and self.wT.get(wo) is not None and \
                self._one_pass_bwd(causal, rows, B, Sq, Sk, q[0].shape[0] if rows is not None else B * Sq):
            ldr = q[0].shape[0] if rows is not None else B * Sq           # = the q_rows the attention call reports: row stride of the [H][rows] table
            assert ldr >= T
            delta_rows = self._delta_rows(ldr)
            self._dgrad(gb, wo, gC, T, d, d, False, rowdot=(ctx, delta_rows, ldr))
        else:
            self._dgrad(gb, wo, gC, T, d, d, False)
        return self._attn_bwd((gC, 0, d), q, k, v, dq, dk, dv, B, Sq, Sk, attn_save, out=(ctx, 0, d), key_mask=key_mask, causal=causal, dbias=dbias,
                              rows=rows, delta_rows=delta_rows)

    def _delta_rows(self, T):
        buf = getattr(self, '_drows', None)
        if buf is None or buf.numel() < self.H * T:
            buf = self._drows = torch.empty(self.H * T, dtype=torch.float32, device=self.device)
        return buf[:self.H * T]

    def backward(self, gy_dec, gy_enc_extra=None):
        """gy_dec: grad wrt decoder output (T,d) storage dtype (None for encoder-only). Writes all parameter gradients
        into the flat buffer currently selected by self._gsel (vector region zeroed first)."""
        sv = self._saved
        if sv is None:
            raise PBError('backward called without a saved forward')
        self._ensure_wT()
        ws = self._cur_ws
        B, S, T, d = sv['B'], sv['S'], ws['T'], self.d
        Te, Td = ws['Te'], ws['Td']                         # rows on the encoder / decoder side (T each unless the step is packed)
        pack = sv.get('pack')
        r_enc, r_dec, r_cross = (pack.enc, pack.dec, pack.cross) if pack is not None else (None, None, None)
        ids_e, ids_d = (pack.src_e, pack.src_d) if pack is not None else (None, None)
        sub = pack.sub if pack is not None else None
        p, seed, wf, g = sv['p'], sv['seed'], self.wf, self.g
        emask, dmask = sv['emask'], sv['dmask']
        gy, galt = (t[:Td] for t in ws['gy'])
        genc = ws['genc'][:Te]
        onehot_route = (self.code == PB_BF16 or (self.x3 and _X3_ONEHOT)) and Te % 64 == 0 and Td % 64 == 0      # dP = Onehot^T dz on the matrix cores, no atomics (bf16x3: dz as two bf16 planes, two GEMMs)
        dec_tab_done = None                                  # event: the decoder tokens' half of dP has been written
        base = ws.get('_base', ws)
        if not _NO_DEFER:
            # ~160 bias / LayerNorm-parameter reductions per pass: keep their partial rows and sum them in one launch at the end
            if 'defer' not in base:
                H, hd, ff = self.H, self.hd, max(self.fe, self.fd)
                per_layer = 3 * self.partials.numel() + int(LIB.query('pb_gemm_colsum_ws_floats', T, ff)) + \
                    2 * int(LIB.query('pb_flash_bias_ws_floats', B, H, S, S, hd if hd in (64, 96, 128) else 64)) + 64
                base['defer'] = (torch.empty((self.NE + self.ND) * per_layer, dtype=torch.float32, device=self.device),
                               torch.empty(64 * (8 * (self.NE + self.ND) + 8), dtype=torch.uint8, device=self.device))
            ops.defer_begin(*base['defer'])
        if gy_dec is not None:
            cur = gy if gy_dec.data_ptr() == gy.data_ptr() else gy_dec
            for l in reversed(range(self.ND)):
                L, pf = ws['dec'][l], 'dec.%d.' % l
                x_in = ws['dec'][l - 1]['y2'] if l > 0 else ws['x_dec']
                # query side of the layer: all decoder rows, or (last layer of a packed step) the rows with a loss term
                Lq, y1, Tq, ids_q, r_x = L, L['y1'], Td, ids_d, r_cross
                if sub is not None and l == self.ND - 1:
                    Lq, Tq, ids_q, r_x = sv['sub_layer'], sub.T, sub.src, sub.cross
                    y1 = Lq['y1s']
                    cur = cur[:Tq]
                gA = self._ffn_ln_bwd(Lq, pf, self.fd, cur, Lq['yc'], seed, self._site('dec', l, 2), p, Tq, row_ids=ids_q)
                # cross-attention block: y_c = LN(y1 + drop(out_c(attn(q_c(y1), kv_c(enc)))))
                dq_d, dqkv_d = self._ring('dq')[:Tq], self._ring('dqkv')[:Td]
                kvld, dkv_all = self.ND * 2 * d, ws['dkv_all']
                if l == self.ND - 1:
                    self._before_write(dkv_all)
                g1 = gy if cur.data_ptr() != gy.data_ptr() else galt
                g1q = g1[:Tq]
                fused = self._attn_block_bwd(Lq, pf, (pf + 'wo_c', pf + 'bo_c'), gA, y1, (Lq['qc'], 0, d), (ws['kvc_all'], l * 2 * d, kvld), (ws['kvc_all'], l * 2 * d + d, kvld),
                                             (dq_d, 0, d), (dkv_all, l * 2 * d, kvld), (dkv_all, l * 2 * d + d, kvld), Lq['ctxc'], Lq['ac'], L['attnc'],
                                             Lq['mc'], Lq['rc'], wf[pf + 'lnc.w'], g[pf + 'lnc.b'], g[pf + 'lnc.w'], g1q, seed, self._site('dec', l, 1), p, B, S, S, emask, False,
                                             dbias=(g[pf + 'bq_c'], g[pf + 'bkv_c'][:d], g[pf + 'bkv_c'][d:]), T=Tq, rows=r_x, row_ids=ids_q)
                if not fused:
                    ops.colsum(dq_d, g[pf + 'bq_c'], self.partials, Tq, d)
                    ops.colsum(dkv_all[:, l * 2 * d:(l + 1) * 2 * d], g[pf + 'bkv_c'], self.partials, Te, 2 * d, ld=kvld)
                self._wgrad(dq_d, y1, pf + 'wq_c', d, d, Tq)
                self._dgrad(dq_d, pf + 'wq_c', g1q, Tq, d, d, True)
                if l == 0:
                    # the key / value side of every layer's cross-attention, in one go: nothing before the encoder's backward reads it
                    self._wgrad(dkv_all, sv['enc_out'], 'dec.wkv_all', kvld, d, Te)
                    self._dgrad(dkv_all, 'dec.wkv_all', genc, Te, d, kvld, False)
                if Tq != Td:
                    # the gradient wrt y1 lives on the loss rows: spread it over the decoder rows (zeros elsewhere) in the buffer `cur` has left
                    full = gy if g1.data_ptr() != gy.data_ptr() else galt
                    self._before_write(full)
                    ops.fill_f32(full.view(torch.float32) if full.dtype != torch.float32 else full, 0.0)
                    ops.scatter_rows16(g1q, sub.idx, full, Tq, d * full.element_size())
                    g1 = full
                # self-attention block
                g2 = gy if g1.data_ptr() != gy.data_ptr() else galt
                bq = g[pf + 'bqkv']
                fused = self._attn_block_bwd(L, pf, (pf + 'wo', pf + 'bo'), g1, x_in, (L['qkv'], 0, 3 * d), (L['qkv'], d, 3 * d), (L['qkv'], 2 * d, 3 * d),
                                             (dqkv_d, 0, 3 * d), (dqkv_d, d, 3 * d), (dqkv_d, 2 * d, 3 * d), L['ctx'], L['a1'], L['attn'],
                                             L['m1'], L['r1'], wf[pf + 'ln1.w'], g[pf + 'ln1.b'], g[pf + 'ln1.w'], g2, seed, self._site('dec', l, 0), p, B, S, S, dmask, True,
                                             dbias=(bq[:d], bq[d:2 * d], bq[2 * d:]), T=Td, rows=r_dec, row_ids=ids_d)
                if not fused:
                    ops.colsum(dqkv_d, bq, self.partials, Td, 3 * d)
                self._wgrad(dqkv_d, x_in, pf + 'wqkv', 3 * d, d, Td)
                self._dgrad(dqkv_d, pf + 'wqkv', g2, Td, d, 3 * d, True)
                cur = g2
                self._ready(pf + 'wqkv', pf + 'w2')
            if self.ND:
                self._ready('dec.wkv_all')
            if sv.get('alt'):
                gpre = cur
                if p > 0.0:
                    gpre = gy if cur is not gy else galt
                    ops.dropout(cur, gpre, seed, self._site('dec_emb'), p)
                dz = ws.setdefault('alt_dz', torch.empty(T, d, dtype=self.xdt, device=self.device))
                scratch = ws.setdefault('alt_db', torch.zeros(d, dtype=torch.float32, device=self.device))
                ops.add_ln_bwd(gpre, ws['alt_pos'], ws['alt_e'], wf['dec.lne.w'], ws['md'], ws['rd'], dz, None, g['dec.lne.w'], g['dec.lne.b'],
                               scratch, self.partials, False, 0, 0, 0.0)
                ops.batch_sum(dz, g['dec.pos'][2:2 + S], B, S * d)
                self._alt_grad = dz                                       # gradient wrt the supplied decoder_inputs_embeds
            else:
                ops.embed_ln_bwd(cur, sv['dec16'], self.ptab, wf['lin.b'], wf['dec.pos'], wf['dec.lne.w'], ws['md'], ws['rd'], self.dptab,
                                 g['dec.pos'], g['lin.b'], g['dec.lne.w'], g['dec.lne.b'], self.partials, S, seed, self._site('dec_emb'), p,
                                 dz_out=ws['dz'][Te:Te + Td] if onehot_route else None, padded=True, row_ids=pack.src_d if pack is not None else None)
                if onehot_route and pack is not None:
                    ops.pos_grad_packed(ws['dz'][Te:Te + Td], pack.inv_d, g['dec.pos'][2:2 + S], B, S)
                elif onehot_route:
                    ops.batch_sum(ws['dz'][T:], g['dec.pos'][2:2 + S], B, S * d)
                if onehot_route:
                    # the decoder tokens' half of dP = Onehot^T dz is ready now: on the second stream, beside the encoder's backward
                    ops.onehot_build(sv['dec16'], ws['onehot'][Te:Te + Td], padded=True)
                    dec_tab_done = self._onehot_gemm(ws['onehot'][Te:Te + Td], ws['dz'][Te:Te + Td], Td, False, side=bool(_SIDE_TAIL))
            cur = genc
            if gy_enc_extra is not None:
                cur = genc.add_(gy_enc_extra)
        else:
            cur = gy_enc_extra
        gy, galt = (t[:Te] for t in ws['gy'])
        for l in reversed(range(self.NE)):
            L, pf = ws['enc'][l], 'enc.%d.' % l
            x_in = ws['enc'][l - 1]['y2'] if l > 0 else ws['x_enc']
            gA = self._ffn_ln_bwd(L, pf, self.fe, cur, L['y1'], seed, self._site('enc', l, 1), p, Te, row_ids=ids_e)
            dqkv_e = self._ring('dqkv')[:Te]
            g2 = gy if cur is not gy else galt
            bq = g[pf + 'bqkv']
            fused = self._attn_block_bwd(L, pf, (pf + 'wo', pf + 'bo'), gA, x_in, (L['qkv'], 0, 3 * d), (L['qkv'], d, 3 * d), (L['qkv'], 2 * d, 3 * d),
                                         (dqkv_e, 0, 3 * d), (dqkv_e, d, 3 * d), (dqkv_e, 2 * d, 3 * d), L['ctx'], L['a1'], L['attn'],
                                         L['m1'], L['r1'], wf[pf + 'ln1.w'], g[pf + 'ln1.b'], g[pf + 'ln1.w'], g2, seed, self._site('enc', l, 0), p, B, S, S, emask, False,
                                         dbias=(bq[:d], bq[d:2 * d], bq[2 * d:]), T=Te, rows=r_enc, row_ids=ids_e)
            if not fused:
                ops.colsum(dqkv_e, bq, self.partials, Te, 3 * d)
            self._wgrad(dqkv_e, x_in, pf + 'wqkv', 3 * d, d, Te)
            self._dgrad(dqkv_e, pf + 'wqkv', g2, Te, d, 3 * d, True)
            cur = g2
            self._ready(pf + 'wqkv', pf + 'w2')
        # the layers have left their last bias / LayerNorm-parameter partial sums: the one launch that reduces them all runs on the second
        # stream, beside the embedding gradients
        side_tail = _SIDE_TAIL and not _NO_DEFER and (_WGRAD_STREAM & 1) and self._side_stream() is not None and self.grad_hook is None
        if side_tail:
            self._event().wait_on(self._side)
            with torch.cuda.stream(self._side):
                ops.defer_flush()
                self._side_last = self._event()
        ops.embed_ln_bwd(cur, sv['enc16'], self.ptab, wf['lin.b'], wf['enc.pos'], wf['enc.lne.w'], ws['me'], ws['re'], self.dptab,
                         g['enc.pos'], g['lin.b'], g['enc.lne.w'], g['enc.lne.b'], self.partials, S, seed, self._site('enc_emb'), p,
                         dz_out=ws['dz'][:Te] if onehot_route else None, padded=True, row_ids=pack.src_e if pack is not None else None)
        if onehot_route:
            # dP = Onehot^T dz over the encoder AND decoder tokens in one split-K MFMA GEMM (K = Te + Td): no atomics
            if pack is not None:
                ops.pos_grad_packed(ws['dz'][:Te], pack.inv_e, g['enc.pos'][2:2 + S], B, S)
            else:
                ops.batch_sum(ws['dz'][:T], g['enc.pos'][2:2 + S], B, S * d)
            ops.onehot_build(sv['enc16'], ws['onehot'][:Te], padded=True)
            # the encoder tokens' half, added to the decoder's (a caller-supplied decoder embedding has no Octuple rows to scatter into)
            if dec_tab_done is not None:
                dec_tab_done.wait_on(torch.cuda.current_stream())
            self._onehot_gemm(ws['onehot'][:Te], ws['dz'][:Te], Te, dec_tab_done is not None, side=False)
        # projected-table gradient -> embedding tables and the shared merge Linear (exact f32)
        E, W, R = self.wf['emb'], self.wf['lin.w'], ops.TAB_ROWS
        lin_w = lambda: ops.gemm(self.dptab, E, g['lin.w'], M=d, N=256, K=R, dtype=PB_F32, a_kc=False, b_kc=False, lda=d, ldb=256, ldc=2048, alpha=16.0,
                                 c_f32=True, nb1=8, sA=(R * d, 0), sB=(R * 256, 0), sC=(256, 0))
        if side_tail:                                           # two small f32 GEMMs that both read dP: one per stream
            self._event().wait_on(self._side)
            with torch.cuda.stream(self._side):
                lin_w()
                self._side_last = self._event()
        ops.gemm(self.dptab, W, g['emb'], M=R, N=256, K=d, dtype=PB_F32, b_kc=False, lda=d, ldb=2048, ldc=256, alpha=16.0, c_f32=True,
                 nb1=8, sA=(R * d, 0), sB=(256, 0), sC=(R * 256, 0))
        if not side_tail:
            lin_w()
        self._ready('emb', 'lin.w')
        if not _NO_DEFER and not side_tail:
            ops.defer_flush()
        self._join_side()
        if self.grad_hook is not None and self.Gcur is self.G32:
            self.grad_hook(self.n_matrix, self.n_total)          # vectors / position tables (accumulated region)

    def _onehot_gemm(self, onehot, dz, K, accum, side):
        """dP (+)= Onehot^T dz over K token rows: one split-K MFMA GEMM (no atomics). side: on the second stream; returns its event."""
        d = self.d
        which = 1 if side else 0
        need = 16 * ops.TAB_TOTAL * d
        if self._slabs_oh[which] is None:
            self._slabs_oh[which] = torch.empty(need, dtype=torch.float32, device=self.device)
        one = lambda b, acc: ops.gemm(onehot, b, self.dptab, M=ops.TAB_TOTAL, N=d, K=K, dtype=PB_BF16, a_kc=False, b_kc=False, lda=ops.TAB_TOTAL, ldb=d,
                                      dbg=self._bwd_dbg(), ldc=d, c_f32=True, accum=acc, splitk=16, slabs=self._slabs_oh[which], tile256=True)
        if self.x3:
            # the one-hot matrix is exact in bf16, so the split-bf16 product is Onehot^T dz_hi + Onehot^T dz_lo: two bf16 GEMMs over dz's two planes (the exact-f32
            # instantiation scatters with f32 atomics: 1.4 ms per side at B = 16 and a summation order that changes from run to run)
            planes = self._cur_ws.get('_base', self._cur_ws)['dz_planes']
            r0 = (dz.data_ptr() - self._cur_ws.get('_base', self._cur_ws)['dz'].data_ptr()) // (4 * d)
            hi, lo = planes[0, r0:r0 + K], planes[1, r0:r0 + K]

            def launch():
                ops.split_bf16(dz[:K], hi, lo)
                one(hi, accum)
                one(lo, True)
        else:
            launch = lambda: one(dz, accum)
        if side and (_WGRAD_STREAM & 1) and self._side_stream() is not None:
            self._event().wait_on(self._side)
            with torch.cuda.stream(self._side):
                launch()
                done = self._event()
            self._side_last = done
            return done
        launch()
        return self._event()

    def zero_accumulated_grads(self):
        """Vector/table gradients are accumulated by the kernels (+=): zero them (and dP) before a backward."""
        ops.fill_f32(self.Gcur[self.n_matrix:], 0.0)
        ops.fill_f32(self.dptab, 0.0)

    def heads_backward(self, dlogits, dec_hidden):
        """dlogits (T,1280) storage dtype -> head grads + grad wrt decoder hidden (returned in ws['gy'][0])."""
        ws, g, T, d = self._cur_ws, self.g, dlogits.shape[0], self.d
        self._ensure_wT()
        ops.colsum(dlogits, g['head.b'], self.partials, T, ops.VOCAB)
        self._wgrad(dlogits, dec_hidden, 'head.w', ops.VOCAB, d, T)
        gy = ws['gy'][0][:T]
        self._dgrad(dlogits, 'head.w', gy, T, d, ops.VOCAB, False)
        self._ready('head.w')
        return gy

    # ------------------------------------------------------------------ gradient buffer selection
    def _select_grads(self, alt):
        if alt and self.G32_alt is None:
            self.G32_alt = torch.zeros_like(self.G32)
        self.Gcur = self.G32_alt if alt else self.G32
        self.g = {}
        for name, s in self.slots.items():
            self.g[name] = self.Gcur[s.off:s.off + s.numel].view(s.shape)
        self._alias_cross_kv(self.g)

    def grad_views_of(self, buf):
        return [buf[self._elem_off(s, r):self._elem_off(s, r) + p.numel()].view(p.shape) for p, (s, r) in zip(self.params, self.param_slots)]

    def optimizer_state(self, model):
        """The AdamW moments keyed by `model`'s parameter names (CPU copies), not as the flat buffers: the order of the slots in the
        flat layout is an implementation detail that has changed between commits, parameter names have not."""
        self.finish_updates()
        out = {'step': self.step_count, 'exp_avg': None, 'exp_avg_sq': None}
        if self.opt_m is not None:
            by_id = {id(p): i for i, p in enumerate(self.params)}
            mv, vv = self.grad_views_of(self.opt_m), self.grad_views_of(self.opt_v)
            names = [(k, by_id[id(p)]) for k, p in model.named_parameters() if id(p) in by_id]
            out['exp_avg'] = {k: mv[i].detach().cpu().clone() for k, i in names}
            out['exp_avg_sq'] = {k: vv[i].detach().cpu().clone() for k, i in names}
        return out

    def load_optimizer_state(self, model, state):
        """Inverse of optimizer_state; refuses a state whose parameter names or shapes do not match this model."""
        self.finish_updates()
        self.step_count = int(state.get('step', 0))
        if state.get('exp_avg') is None:
            self.opt_m = self.opt_v = None
            return
        if not isinstance(state['exp_avg'], dict):
            raise PBError('optimizer state holds flat moment buffers without a layout (written before round 5): they cannot be matched to the '
                          'parameters safely; resume without them')
        if self.opt_m is None:
            self.opt_m = torch.zeros_like(self.P32); self.opt_v = torch.zeros_like(self.P32)
        by_id = {id(p): i for i, p in enumerate(self.params)}
        mv, vv = self.grad_views_of(self.opt_m), self.grad_views_of(self.opt_v)
        names = {k: by_id[id(p)] for k, p in model.named_parameters() if id(p) in by_id}
        if set(names) != set(state['exp_avg']) or set(names) != set(state['exp_avg_sq']):
            raise PBError('optimizer state does not match the model: %d parameters here, %d in the state' % (len(names), len(state['exp_avg'])))
        with torch.no_grad():
            for k, i in names.items():
                if tuple(state['exp_avg'][k].shape) != tuple(mv[i].shape):
                    raise PBError('optimizer state: shape of %s is %s, the model has %s' % (k, tuple(state['exp_avg'][k].shape), tuple(mv[i].shape)))
                mv[i].copy_(state['exp_avg'][k]); vv[i].copy_(state['exp_avg_sq'][k])

    # ------------------------------------------------------------------ module-level (autograd) entry points
    def _prep_inputs(self, enc_ids, dec_ids, emask, dmask):
        if enc_ids.device != self.device:
            raise PBError('inputs are on %s but the model is on %s' % (enc_ids.device, self.device))
        enc16 = ops.ids_to_i16(enc_ids)
        dec16 = ops.ids_to_i16(dec_ids) if dec_ids is not None else None
        self.note_ids(enc16)
        if dec16 is not None and dec16.dim() == 3:
            self.note_ids(dec16)
        f = lambda m: None if m is None else m.to(dtype=torch.float32).contiguous()
        return enc16, dec16, f(emask), f(dmask)

    def note_ids(self, ids16, owned=True):
        """Enqueue the range check of (..., 8) Octuple ids (PianoBart.py:15-16: nn.Embedding raises IndexError on an id outside its
        table); the verdict is read by check_ids() at a point where the host waits for the device anyway. The kernel rewrites an
        offending id to 0 so that the gathers queued behind it stay inside their tables -- in a tensor the ENGINE owns: `owned=False`
        (a caller's int16 tensor) is checked on a private copy, which is returned and must be the one the step reads; the caller's
        tensor is never written."""
        if not owned:
            ids16 = ids16.clone()
        if getattr(self, '_id_flag', None) is None or self._id_flag.device != ids16.device:
            self._id_flag = torch.zeros(1, dtype=torch.int32, device=ids16.device)
            self._id_lim = torch.tensor(ops.SEG_SIZES, dtype=torch.int32, device=ids16.device)
        ops.ids_check(ids16, self._id_lim, self._id_flag)
        return ids16

    def check_ids(self, collective=True):
        """Synchronises. Raises IndexError if a checked batch held an id outside its embedding table. collective=True (the module route's
        TRAINING forward, which every rank of a torchrun job runs in step): the flag is max-reduced first, so the error is raised on EVERY rank
        when any rank's batch tripped it (a rank that raised alone would leave the others waiting in the next gradient exchange).
        collective=False (`generate` and the eval forwards: calls that a rank may make by itself -- rank-0 validation, a demo): local verdict
        only, no exchange that the other ranks would have to join. The fused step does not come through here: its verdict is
        local (`_raise_if_bad_ids`) and the trainers validate the loader's host batch on every rank before the copy."""
        self._id_verdicts = []
        if collective and getattr(self, '_id_flag', None) is not None and self.grad_hook is not None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.all_reduce(self._id_flag, op=dist.ReduceOp.MAX)
        if getattr(self, '_id_flag', None) is not None and int(self._id_flag.item()) != 0:
            self._id_flag.zero_()
            raise IndexError('index out of range in self: an Octuple id lies outside its embedding table (sizes %s)' % ops.SEG_SIZES)

    def _queue_id_verdict(self):
        """The fused step does not drain the stream: the mark of note_ids travels to pinned memory behind an event and is read by
        _raise_if_bad_ids wherever the host already knows the event has passed (pack_batch's wait in the same call; else the next call)."""
        pool = getattr(self, '_id_pins', None)
        if pool is None:
            pool = self._id_pins = [[torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(4)], 0]
            self._id_verdicts = []
        pool[1] = (pool[1] + 1) % len(pool[0])
        pin = pool[0][pool[1]]
        pin.copy_(self._id_flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._id_verdicts.append((pin, ev))
        del self._id_verdicts[:-3]

    def _raise_if_bad_ids(self, wait=False):
        """wait: block until every queued verdict has landed (optimizer_step: nn.Embedding raises BEFORE any update, PianoBart.py:15-16, so the
        update must not be enqueued for a batch whose verdict is still out; the range check is the first kernel of its step, the wait ends
        long before the step does, and only callers that hand over unchecked ids ever queue one)."""
        q = getattr(self, '_id_verdicts', None)
        while q and (wait or q[0][1].query()):
            if wait:
                q[0][1].synchronize()
            pin, _ = q.pop(0)
            if int(pin[0]) != 0:
                q.clear()
                self._id_flag.zero_()
                raise IndexError('index out of range in self: an Octuple id lies outside its embedding table (sizes %s)' % ops.SEG_SIZES)

    def _next_seed(self):
        self._seed = (self._seed * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return self._seed

    def module_forward_logits(self, enc_ids, dec_ids, emask, dmask, training):
        if self.mlm is None:
            raise PBError('engine has no LM heads')
        if enc_ids.device.type != 'cuda':
            raise PBError('pianobart_amd needs HIP device tensors (got %s); there is no CPU path' % enc_ids.device)
        self.bind(enc_ids.device)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.params)
        out = _LMFn.apply(self, enc_ids, dec_ids, emask, dmask, training, need_grad, *self.params)
        self.check_ids(collective=bool(training))                        # nn.Embedding's IndexError (the module route hands tensors to host code next anyway)
        return out

    def module_forward_hidden(self, enc_ids, dec_ids, emask, dmask, training, dec_embeds=None):
        if enc_ids.device.type != 'cuda':
            raise PBError('pianobart_amd needs HIP device tensors (got %s); there is no CPU path' % enc_ids.device)
        self.bind(enc_ids.device)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.params)
        n_backbone = len(self.params) - (16 if self.mlm is not None else 0)
        out = _HiddenFn.apply(self, enc_ids, dec_ids, emask, dmask, training, need_grad, dec_embeds, *self.params[:n_backbone])
        self.check_ids(collective=bool(training))
        return out

    # ------------------------------------------------------------------ fused pre-train step (bench / Pretrainer)
    def loss_and_grads(self, enc16, dec16, tgt16, loss_mask, emask, dmask, train=True, count_hook=None, head_w=None, w_scale=1.0,
                       argmax_out=None, ids_checked=False):
        """Forward + fused CE/argmax/acc + full backward. Returns the (24,) device tensor of sums
        {sum ce*m, sum m, sum correct*m} x 8 heads. `count_hook(counts)` may all-reduce the 8 mask counts (DP).
        ids_checked: the caller vouches that every id lies inside its embedding table (ids generated on the device, or a host batch it
        has validated); otherwise the ids are range-checked here (PianoBart.py:15-16: nn.Embedding raises IndexError; an offending id is
        replaced by 0 so that no gather leaves its table) and IndexError is raised at the first point where the host knows the verdict
        without draining the stream: inside this call when the packing plan waits for its row counts, else on the next engine call."""
        B, S = enc16.shape[:2]
        T = B * S
        self._raise_if_bad_ids()
        if not ids_checked:
            enc16 = self.note_ids(enc16, owned=False)
            dec16 = self.note_ids(dec16, owned=False)
            self._queue_id_verdict()
        seed = self._next_seed()
        self._select_grads(False)
        self._tables_ready = False
        pack = self._pack_batch(enc16, dec16, tgt16, loss_mask, emask, dmask) if (_PACK_ROWS and argmax_out is None) else None
        self._raise_if_bad_ids()
        if pack is not None:
            enc16, dec16, tgt, lm = pack.enc16, pack.dec16, pack.tgt16, pack.loss_mask
            if pack.sub is not None:                                   # the last decoder layer hands over the rows with a loss term only
                tgt, lm = pack.sub.tgt16, pack.sub.loss_mask
        else:
            tgt, lm = tgt16.reshape(T, 8), loss_mask.reshape(T, 8)
        self.last_rows = (pack.Te, pack.Td, T, pack.sub.T if pack.sub is not None else pack.Td) if pack is not None else (T, T, T, T)
        # (query, key) pairs the three attention forms really cover (bench.py prices the step with them)
        self.last_pairs = pack.pairs if pack is not None else (B * S * S, B * S * S // 2, B * S * S)
        dec_h, _ = self.forward_hidden(enc16, dec16, emask, dmask, train, seed, pack=pack)
        logits = self.heads_forward(dec_h)
        ws = self._cur_ws
        sums, counts, coef = self.scal[0:24], self.scal[24:32], self.scal[32:40]
        ops.fill_f32(sums, 0.0)
        ops.mask_count(lm, counts, self.partials)
        if count_hook is not None:
            count_hook(counts)
        ops.loss_coef(counts, self.loss_w if head_w is None else head_w, coef, w_scale)
        dlogits = ws['dlogits'][:logits.shape[0]] if train else None
        ops.ce_fwd_bwd(logits, tgt, lm, sums, self.partials, coef, dlogits, argmax_out)
        if train:
            self.zero_accumulated_grads()
            gy = self.heads_backward(dlogits, dec_h)
            self.backward(gy)
        return sums

    def _pack_batch(self, enc16, dec16, tgt16, loss_mask, emask, dmask):
        return rowpack.pack_batch(self, enc16, dec16, tgt16, loss_mask, emask, dmask)

    def prefetch_pack(self, loss_mask, emask, dmask, stream=None):
        """Optional pipeline hint: the batch with these masks is the NEXT one handed to loss_and_grads (rowpack.prefetch_counts)."""
        if _PACK_ROWS:
            rowpack.prefetch_counts(self, loss_mask, emask, dmask, stream)

    def optimizer_step(self, lr=2e-5, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01, max_norm=3.0, gscale=1.0):
        """clip_grad_norm_(3.0) + HF AdamW on the flat buffers, refreshing the bf16 shadow (pretrain.py:195-196)."""
        self._raise_if_bad_ids(wait=True)
        if self.opt_m is None:
            self.opt_m = torch.zeros_like(self.P32)
            self.opt_v = torch.zeros_like(self.P32)
        self.step_count += 1
        sq, clip = self.scal[40:41], self.scal[41:42]
        ops.grad_sqnorm(self.G32, self.partials, sq)
        ops.clip_coef(sq, max_norm, gscale, clip)
        if self.pipeline_updates and self.code == PB_BF16 and _WGRAD_STREAM and self._side_stream() is not None:
            # The update itself streams 30 bytes per parameter (1.3 ms at cfg 2) and only the NEXT forward needs its result, layer by
            # layer: run it on the second stream in the order the forward reads the parameters, in three groups with an event each
            # (_await_updates), followed by the transposed copies the backward reads.
            self._event().wait_on(self._side)
            with torch.cuda.stream(self._side):
                evs = []
                for ranges in self._update_groups():
                    for lo, hi in ranges:
                        ops.adamw_step(self.P32[lo:hi], self.G32[lo:hi], self.opt_m[lo:hi], self.opt_v[lo:hi], self.Pbf[lo:hi], clip, lr, betas[0], betas[1],
                                       eps, weight_decay, self.step_count)
                    evs.append(self._event())
                self._versions = sum(p._version for p in self.params)
                self._shadow_gen += 1
                if self._wT_table is not None:
                    ops.transpose_batch_bf16(self.Pbf, self.PbfT, self._wT_table, self._wT_tiles)
                    self._wT_done = self._event()
                    self._wT_gen = self._shadow_gen
            self._upd, self._upd_waited = evs, 0
            return
        ops.adamw_step(self.P32, self.G32, self.opt_m, self.opt_v, self.Pbf, clip, lr, betas[0], betas[1], eps, weight_decay, self.step_count)
        if self.code == PB_BF16:
            self._versions = sum(p._version for p in self.params)
            self._shadow_gen += 1
            self._refresh_wT(on_side=True)

    def _update_groups(self):
        """Flat-buffer ranges of the three update groups: {vectors and tables, Octuple embeddings, the first two encoder layers},
        {the other encoder layers}, {decoder layers, LM heads}."""
        if self._upd_groups is None:
            first = lambda names: next((self.slots[n].off for n in names if n in self.slots), self.n_matrix)
            a2 = first(['dec.0.wqkv', 'head.w'])
            a1 = min(first(['enc.2.wqkv']), a2)
            self._upd_groups = [[(0, a1), (self.n_matrix, self.n_total)], [(a1, a2)], [(a2, self.n_matrix)]]
            self._upd_groups = [[(lo, hi) for lo, hi in g if hi > lo] for g in self._upd_groups]
        return self._upd_groups

    def _await_updates(self, group=2):
        """The current stream waits until the pipelined optimizer step has written update groups 0 .. group."""
        if self._upd is None:
            return
        cur = torch.cuda.current_stream()
        while self._upd_waited <= group:
            self._upd[self._upd_waited].wait_on(cur)
            self._upd_waited += 1
        if self._upd_waited >= len(self._upd):
            self._upd = None

    def finish_updates(self):
        """Call before reading parameters from outside the engine (checkpoints) when pipeline_updates is on."""
        self._await_updates(2)

    # ------------------------------------------------------------------ generate (model.py:28-66)
    def generate(self, enc_ids, emask, sample_row, use_cache=True, max_new=None, sampler=None):
        """Autoregressive decode with the reference's control flow (SOS start, host-side nucleus sampling, early stop on
        any special token). The reference re-runs encoder AND decoder over all S positions for every generated position
        (model.py:42-45); here the encoder runs once, the cross-attention K/V of every decoder layer are projected once,
        and each step feeds ONE decoder token through the layers against a self-attention K/V cache. Position-i logits
        only depend on decoder inputs <= i (causal), so the tokens are identical (tests/test_model_gpu.py).
        max_new: stop after that many positions (None = the window). sampler = dict(T=[8 temperatures], P=[8 thresholds]): the caller
        states that `sample_row` IS model.py:68-107 with these constants, drawing np.random.random_sample(8) per position; the decoder may
        then sample on the device ahead of the host (`_generate_device_sampled`) -- `sample_row` still decides every token."""
        self._await_updates(2)
        if not use_cache:
            return self._generate_nocache(enc_ids, emask, sample_row)
        if self.hd not in (32, 64, 96, 128):                 # pb_attn_decode's row-chunk layouts; other head sizes use the training kernels
            return self._generate_pyloop(enc_ids, emask, sample_row)
        import ctypes
        from ._lib import DecodePlan
        pb, d, X = self.pb, self.d, self.xdt
        self.bind(enc_ids.device)
        S, dev = enc_ids.shape[1], enc_ids.device
        pad = torch.from_numpy(pb.pad_word_np).to(dev)
        pad_cpu = torch.from_numpy(pb.pad_word_np)
        result = pad.repeat(1, S, 1)
        em = emask.to(torch.float32).contiguous() if emask is not None else None
        enc16 = ops.ids_to_i16(enc_ids)
        self.note_ids(enc16); self.check_ids(collective=False)
        e = lambda *shape, dt=X: torch.empty(*shape, dtype=dt, device=dev)
        with torch.no_grad():
            _, enc_out = self.forward_hidden(enc16, None, em, None, False, 0)
            wf, ff = self.wf, self.fd
            kvc = [e(S, 2 * d) for _ in range(self.ND)]
            for l in range(self.ND):
                self._linear(enc_out, 'dec.%d.wkv_c' % l, 'dec.%d.bkv_c' % l, kvc[l], S, 2 * d, d)
            kvs = [torch.zeros(S, 2 * d, dtype=X, device=dev) for _ in range(self.ND)]
            rows = {n: e(1, d) for n in ('x', 'y1', 'yc', 'y2', 'q', 'ctx', 'a')}
            rows['g'] = e(1, ff)
            stat = torch.empty(8, dtype=torch.float32, device=dev)
            logits = torch.empty(1, ops.VOCAB, dtype=torch.float32, device=dev)
            tok16 = torch.tensor(pb.sos_word_np, dtype=torch.int16, device=dev)
            plan = DecodePlan()
            s_enc = S
            if em is not None:                                               # keys behind the last visible encoder position are masked for every query: stop there
                km = torch.empty(1, dtype=torch.int32, device=dev)
                ops.key_extent(em, km)
                s_enc = max(1, min(S, int(km.item())))
            plan.dtype, plan.d, plan.H, plan.ffn, plan.S, plan.S_enc, plan.n_layers, plan.vocab = self.code, d, self.H, ff, S, s_enc, self.ND, ops.VOCAB
            for k in range(9):
                plan.tab_off[k] = ops.TAB_OFF[k]
            P = lambda t: t.data_ptr()
            plan.tok16, plan.ptab, plan.lin_b, plan.pos = P(tok16), P(self.ptab), P(wf['lin.b']), P(wf['dec.pos'])
            plan.lne_w, plan.lne_b, plan.enc_mask = P(wf['dec.lne.w']), P(wf['dec.lne.b']), (P(em) if em is not None else None)
            for n, t in rows.items():
                setattr(plan, n, P(t))
            plan.stat, plan.logits, plan.head_w, plan.head_b = P(stat), P(logits), P(self.w['head.w']), P(wf['head.b'])
            attn_part = torch.empty(self.H * 16 * (self.hd + 4), dtype=torch.float32, device=dev)      # PB_DECODE_MAX_SPLITS records per head
            # the split records are merged in the out-projection GEMV's prologue, which holds K = d in one chunk per thread (256 threads x
            # 16 bytes): wider models keep the one-workgroup-per-head attention
            epv = 8 if self.code == PB_BF16 else 4
            plan.attn_part = P(attn_part) if (_DECODE_SPLIT and d <= 256 * epv) else None
            for l in range(self.ND):
                pf, L = 'dec.%d.' % l, plan.layers[l]
                L.wqkv, L.bqkv, L.wo, L.bo = P(self.w[pf + 'wqkv']), P(wf[pf + 'bqkv']), P(self.w[pf + 'wo']), P(wf[pf + 'bo'])
                L.ln1_w, L.ln1_b = P(wf[pf + 'ln1.w']), P(wf[pf + 'ln1.b'])
                L.wq_c, L.bq_c, L.wo_c, L.bo_c = P(self.w[pf + 'wq_c']), P(wf[pf + 'bq_c']), P(self.w[pf + 'wo_c']), P(wf[pf + 'bo_c'])
                L.lnc_w, L.lnc_b = P(wf[pf + 'lnc.w']), P(wf[pf + 'lnc.b'])
                L.w1, L.b1, L.w2, L.b2 = P(self.w[pf + 'w1']), P(wf[pf + 'b1']), P(self.w[pf + 'w2']), P(wf[pf + 'b2'])
                L.ln2_w, L.ln2_b = P(wf[pf + 'ln2.w']), P(wf[pf + 'ln2.b'])
                L.kv_self, L.kv_cross = P(kvs[l]), P(kvc[l])
            pref = ctypes.byref(plan)
            stream = ops._stream()
            res_cpu = pad_cpu.repeat(S, 1)
            # One hipGraph replay per token where the fused decode kernels cover the shape (bf16, head_dim 64 / 128, d a multiple of
            # 256 up to 1024): pb_decoder_* keeps the position in device memory; PB_DECODE_GRAPH=0 issues the same launches directly,
            # PB_DECODE_GRAPH=-1 keeps the round-2 loop below (A/B)
            dec = ctypes.c_void_p()
            self.last_decode = None
            rc_dec = int(LIB.query('pb_decoder_create', pref, ctypes.byref(dec))) if (_DECODE_GRAPH >= 0 and _DECODE_SPLIT) else 1
            if rc_dec < 0:                                              # 1 = the fused kernels do not cover this shape (the loop below does); < 0 is an error
                raise PBError('pb_decoder_create failed (%d): %s' % (rc_dec, LIB.load().pb_last_error().decode()))
            if rc_dec == 0 and sampler is not None and _DECODE_SPEC:
                try:
                    LIB.call('pb_decoder_reset', dec, stream, _DECODE_GRAPH)
                    torch.cuda.current_stream().synchronize()
                    self._generate_device_sampled(dec, sample_row, sampler, S, s_enc, res_cpu, pad_cpu, max_new)
                finally:
                    LIB.call('pb_decoder_destroy', dec)
                return res_cpu.to(dev).unsqueeze(0)
            if rc_dec == 0:
                try:
                    LIB.call('pb_decoder_reset', dec, stream, _DECODE_GRAPH)
                    tok_np = np.asarray(pb.sos_word_np, dtype=np.int16).copy()
                    logit_cpu = torch.empty(ops.VOCAB, dtype=torch.float32)
                    tok_p, log_p = ctypes.c_void_p(tok_np.ctypes.data), ctypes.c_void_p(logit_cpu.data_ptr())
                    n = 0
                    torch.cuda.current_stream().synchronize()           # the prompt's encoder pass: not part of the per-token time below
                    t_loop = time.perf_counter()
                    for i in range(S if max_new is None else min(S, max_new)):
                        LIB.call('pb_decoder_step', dec, tok_p, log_p)
                        n += 1
                        tok = sample_row(logit_cpu)
                        if (tok >= pad_cpu).any():
                            break
                        res_cpu[i] = tok
                        tok_np[:] = tok.numpy()
                    self.last_decode = dict(launches_per_token=int(LIB.query('pb_decoder_launches', dec)), graph=bool(LIB.query('pb_decoder_graph', dec)),
                                            tokens=n, loop_ms=(time.perf_counter() - t_loop) * 1e3, s_enc=s_enc)
                finally:
                    LIB.call('pb_decoder_destroy', dec)
                return res_cpu.to(dev).unsqueeze(0)
            tok_pin = torch.empty(8, dtype=torch.int16).pin_memory()         # one small H2D per position; the result goes up once at the end
            logit_pin = torch.empty(ops.VOCAB, dtype=torch.float32).pin_memory()
            for i in range(S):
                LIB.call('pb_decode_step', pref, i, stream)
                logit_pin.copy_(logits[0])                                  # D2H on the current stream, returns when the row has landed
                tok = sample_row(logit_pin)
                if (tok >= pad_cpu).any():
                    break
                res_cpu[i] = tok
                tok_pin.copy_(tok)
                tok16.copy_(tok_pin, non_blocking=True)                     # stream-ordered before the next step's kernels
            result = res_cpu.to(dev).unsqueeze(0)
        return result

    def _generate_device_sampled(self, dec, sample_row, sampler, S, s_enc, res_cpu, pad_cpu, max_new):
        """The decode loop without a host round trip per token (round 6). The 8 uniform draws of a position do not depend on its logits
        (np.random.choice inside nucleus(), model.py:97), so all S x 8 are drawn AHEAD from a copy of the global RNG state and uploaded;
        the device then samples each position itself (pb_decoder_sampler_init: model.py:68-107 in pb_nucleus_rows' arithmetic order) and
        runs on, 8 tokens per hipGraph replay, two runs in flight. The host follows one run behind: for every position it calls
        `sample_row` on the logged logits row -- the reference code path, consuming the GLOBAL RNG exactly as the per-token loop did, so
        np.random.get_state() ends where the reference's does -- and compares with the ids the device chose. They differ only where the
        device's softmax rounding (1 ulp against torch's CPU softmax) crosses a threshold or a tie; then the decoder is rewound to that
        position with the host's token and everything decoded behind it is discarded. The result is the host's, token for token."""
        import ctypes
        from collections import deque
        K, vocab = 8, ops.VOCAB
        limit = S if max_new is None else max(0, min(S, int(max_new)))
        state = np.random.get_state()
        ahead = np.random.RandomState()
        ahead.set_state(state)
        U = np.ascontiguousarray(ahead.random_sample(S * 8))
        n8 = np.asarray([ops.SEG_OFF[j + 1] - ops.SEG_OFF[j] for j in range(8)], dtype=np.int32)
        off8 = np.asarray(ops.SEG_OFF[:8], dtype=np.int32)
        t8, p8 = np.asarray(sampler['T'], dtype=np.float32), np.asarray(sampler['P'], dtype=np.float32)
        fault = int(getattr(self, 'decode_fault_period', 0) or 0)                # tests: the device's choice is corrupted at every fault-th position
        LIB.call('pb_decoder_sampler_init', dec, t8.ctypes.data, p8.ctypes.data, n8.ctypes.data, off8.ctypes.data, U.ctypes.data, S * 8, fault)
        lp, tp = ctypes.c_void_p(), ctypes.c_void_p()
        LIB.call('pb_decoder_logs', dec, ctypes.byref(lp), ctypes.byref(tp))
        log_logits = torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * (S * vocab)).from_address(lp.value)).reshape(S, vocab))
        log_tok = np.ctypeslib.as_array((ctypes.c_int16 * (S * 8)).from_address(tp.value)).reshape(S, 8)
        first = np.asarray(self.pb.sos_word_np, dtype=np.int16).copy()
        runs, enq, n, rewinds, stop = deque(), 0, 0, 0, False

        def launch(tok=None):
            nonlocal enq
            cnt = min(K, limit - enq)
            tk = int(LIB.query('pb_decoder_launch', dec, cnt, None if tok is None else tok.ctypes.data))
            if tk < 0:
                raise PBError('pb_decoder_launch failed (%d): %s' % (tk, LIB.load().pb_last_error().decode()))
            runs.append((tk, enq, cnt))
            enq += cnt

        t_loop = time.perf_counter()
        if limit > 0:
            launch(first)
        while not stop and (runs or enq < limit):
            while len(runs) < 2 and enq < limit:
                launch()
            tk, start, cnt = runs.popleft()
            LIB.call('pb_decoder_wait', dec, tk)
            for i in range(start, start + cnt):
                tok = sample_row(log_logits[i])
                n += 1
                if (tok >= pad_cpu).any():
                    stop = True
                    break
                res_cpu[i] = tok
                t16 = tok.numpy().astype(np.int16)
                if not np.array_equal(t16, log_tok[i]):                  # the device chose another id here: its later positions are void
                    rewinds += 1
                    LIB.call('pb_decoder_seek', dec, i, t16.ctypes.data)
                    runs.clear()
                    enq = i + 1
                    break
        self.last_decode = dict(launches_per_token=int(LIB.query('pb_decoder_launches', dec)), graph=bool(LIB.query('pb_decoder_graph', dec)),
                                tokens=n, loop_ms=(time.perf_counter() - t_loop) * 1e3, s_enc=s_enc, device_sampler=True, rewinds=rewinds,
                                tokens_per_graph_replay=K)

    def _generate_pyloop(self, enc_ids, emask, sample_row):
        """KV-cached decode sequenced from Python with the training kernels (M = 1 GEMMs, flash attention with one query):
        kept as a cross-check of the native pb_decode_step path."""
        pb, d, H, X = self.pb, self.d, self.H, self.xdt
        self.bind(enc_ids.device)
        S, dev = enc_ids.shape[1], enc_ids.device
        pad = torch.from_numpy(pb.pad_word_np).to(dev)
        pad_cpu = torch.from_numpy(pb.pad_word_np)
        result = pad.repeat(1, S, 1)
        em = emask.to(torch.float32).contiguous() if emask is not None else None
        enc16 = ops.ids_to_i16(enc_ids)
        self.note_ids(enc16); self.check_ids(collective=False)
        e = lambda *shape, dt=X: torch.empty(*shape, dtype=dt, device=dev)
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        with torch.no_grad():
            _, enc_out = self.forward_hidden(enc16, None, em, None, False, 0)
            wf, ff = self.wf, self.fd
            kvc = [e(S, 2 * d) for _ in range(self.ND)]
            for l in range(self.ND):
                self._linear(enc_out, 'dec.%d.wkv_c' % l, 'dec.%d.bkv_c' % l, kvc[l], S, 2 * d, d)
            kvs = [e(S, 2 * d) for _ in range(self.ND)]
            x, q, ctx, a, y1, qc, ctxc, yc, y2 = (e(1, d) for _ in range(9))
            u, g = e(1, ff), e(1, ff)
            mr = f(8)
            logits = f(1, ops.VOCAB)
            save = dict(lse=f(1, H, 1)) if self.use_flash else dict(P=e(1, H, 1, S))
            cur = torch.tensor(pb.sos_word_np, device=dev).reshape(1, 1, 8)
            for i in range(S):
                tok16 = ops.ids_to_i16(cur)
                ops.embed_ln_fwd(tok16.reshape(1, 8), self.ptab, wf['lin.b'], wf['dec.pos'][i:], wf['dec.lne.w'], wf['dec.lne.b'], x,
                                 mr[0:1], mr[1:2], 1, LN_EPS, 0, 0, 0.0, padded=True)
                h = x
                for l in range(self.ND):
                    pf = 'dec.%d.' % l
                    wqkv, bqkv = self.w[pf + 'wqkv'], wf[pf + 'bqkv']
                    ops.gemm(h, wqkv, q, M=1, N=d, K=d, dtype=self.gcode, bias=bqkv[:d])
                    ops.gemm(h, wqkv, kvs[l], M=1, N=2 * d, K=d, dtype=self.gcode, bias=bqkv[d:], b_off=d * d, c_off=i * 2 * d)
                    self._attn_fwd((q, 0, d), (kvs[l], 0, 2 * d), (kvs[l], d, 2 * d), (ctx, 0, d), None, False, 1, 1, i + 1, save)
                    self._linear(ctx, pf + 'wo', pf + 'bo', a, 1, d, d)
                    ops.add_ln_fwd(h, a, wf[pf + 'ln1.w'], wf[pf + 'ln1.b'], y1, mr[2:3], mr[3:4], LN_EPS, 0, 0, 0.0)
                    self._linear(y1, pf + 'wq_c', pf + 'bq_c', qc, 1, d, d)
                    self._attn_fwd((qc, 0, d), (kvc[l], 0, 2 * d), (kvc[l], d, 2 * d), (ctxc, 0, d), em, False, 1, 1, S, save)
                    self._linear(ctxc, pf + 'wo_c', pf + 'bo_c', a, 1, d, d)
                    ops.add_ln_fwd(y1, a, wf[pf + 'lnc.w'], wf[pf + 'lnc.b'], yc, mr[4:5], mr[5:6], LN_EPS, 0, 0, 0.0)
                    self._linear(yc, pf + 'w1', pf + 'b1', g, 1, ff, d, gelu_aux_out=u)
                    self._linear(g, pf + 'w2', pf + 'b2', a, 1, d, ff)
                    out = y2 if h is not y2 else x
                    ops.add_ln_fwd(yc, a, wf[pf + 'ln2.w'], wf[pf + 'ln2.b'], out, mr[6:7], mr[7:8], LN_EPS, 0, 0, 0.0)
                    h = out
                ops.gemm(h, self.w['head.w'], logits, M=1, N=ops.VOCAB, K=d, dtype=self.gcode, bias=wf['head.b'], c_f32=True)
                tok = sample_row(logits[0].cpu())
                if (tok >= pad_cpu).any():
                    break
                result[:, i, :] = tok.to(dev)
                cur = tok.to(dev).reshape(1, 1, 8)
        return result

    def _generate_nocache(self, enc_ids, emask, sample_row):
        """The reference's schedule minus the redundant encoder re-runs: full decoder pass per position (kept as the
        cross-check of the cached path)."""
        pb = self.pb
        self.bind(enc_ids.device)
        S = enc_ids.shape[1]
        dev = enc_ids.device
        pad = torch.from_numpy(pb.pad_word_np).to(dev)
        dec = pad.repeat(1, S, 1)
        result = pad.repeat(1, S, 1)
        dmask = torch.zeros(1, S, dtype=torch.float32, device=dev)
        dec[:, 0, :] = torch.tensor(pb.sos_word_np, device=dev)
        dmask[:, 0] = 1
        pad_cpu = torch.from_numpy(pb.pad_word_np)
        em = emask.to(torch.float32).contiguous() if emask is not None else None
        enc16 = ops.ids_to_i16(enc_ids)
        self.note_ids(enc16); self.check_ids(collective=False)
        with torch.no_grad():
            for i in range(S):
                dec16 = ops.ids_to_i16(dec)
                dec_h, _ = self.forward_hidden(enc16, dec16, em, dmask, False, 0, reuse_encoder=(i > 0))
                logits = self.heads_forward(dec_h)
                cur = sample_row(logits[i].float().cpu())
                if i != S - 1:
                    dec[:, i + 1, :] = cur.to(dev)
                    dmask[:, i + 1] += 1
                if (cur >= pad_cpu).any():
                    break
                result[:, i, :] = cur.to(dev)
        return result


class _LMFn(torch.autograd.Function):
    """PianoBartLM train branch as ONE autograd node around the explicit HIP forward/backward schedules."""

    @staticmethod
    def forward(ctx, eng, enc_ids, dec_ids, emask, dmask, training, need_grad, *params):
        enc16, dec16, em, dm = eng._prep_inputs(enc_ids, dec_ids, emask, dmask)
        B, S = enc16.shape[:2]
        seed = eng._next_seed() if training else 0
        dec_h, _ = eng.forward_hidden(enc16, dec16, em, dm, training, seed)
        logits = eng.heads_forward(dec_h)
        eng._fwd_token += 1
        ctx.eng, ctx.token = eng, eng._fwd_token
        out = logits.view(B, S, ops.VOCAB)
        return out.clone()        # the workspace buffer is overwritten by the next forward

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.eng
        if ctx.token != eng._fwd_token:
            raise PBError('backward for a stale forward: the engine keeps the activations of the most recent forward only')
        ws = eng._cur_ws
        T = ws['T']
        # if .grad already aliases our primary buffer (accumulation without zero_grad), write into the alternate one
        p0 = eng.params[0]
        alias = p0.grad is not None and p0.grad.data_ptr() == eng.grad_views[0].data_ptr()
        eng._select_grads(alias)
        dl = dlogits.reshape(T, ops.VOCAB).contiguous()
        if eng.xdt == torch.float32:
            dlx = dl.float()
        else:
            dlx = ws['dlogits']
            ops.cast_f32_to_bf16(dl.float().contiguous(), dlx)
        eng.zero_accumulated_grads()
        gy = eng.heads_backward(dlx, eng._saved['dec_out'])
        eng.backward(gy)
        grads = eng.grad_views_of(eng.Gcur)
        eng._select_grads(False)
        return (None,) * 7 + tuple(grads)


class _HiddenFn(torch.autograd.Function):
    """PianoBart.forward (hidden states, f32 views for API compatibility)."""

    @staticmethod
    def forward(ctx, eng, enc_ids, dec_ids, emask, dmask, training, need_grad, dec_embeds, *params):
        enc16, dec16, em, dm = eng._prep_inputs(enc_ids, dec_ids, emask, dmask)
        B, S = enc16.shape[:2]
        seed = eng._next_seed() if training else 0
        de = None if dec_embeds is None else dec_embeds.detach().reshape(B * S, eng.d).to(eng.xdt).contiguous()
        dec_h, enc_h = eng.forward_hidden(enc16, dec16, em, dm, training, seed, dec_embeds=de)
        ctx.alt_shape = None if dec_embeds is None else dec_embeds.shape
        if dec16 is None and de is None:
            eng._saved = dict(enc16=enc16, dec16=None, emask=em, dmask=None, p=eng.p_drop if training else 0.0, seed=seed,
                              enc_out=enc_h, dec_out=None, B=B, S=S)
        eng._fwd_token += 1
        ctx.eng, ctx.token, ctx.has_dec, ctx.nparams = eng, eng._fwd_token, (dec16 is not None or de is not None), len(params)
        enc_o = enc_h.float().view(B, S, eng.d).clone() if enc_h.dtype == torch.float32 else enc_h.float().view(B, S, eng.d)
        if not ctx.has_dec:
            return torch.zeros(0, device=enc_o.device), enc_o
        dec_o = dec_h.float().view(B, S, eng.d).clone() if dec_h.dtype == torch.float32 else dec_h.float().view(B, S, eng.d)
        return dec_o, enc_o

    @staticmethod
    def backward(ctx, d_dec, d_enc):
        eng = ctx.eng
        if ctx.token != eng._fwd_token:
            raise PBError('backward for a stale forward: the engine keeps the activations of the most recent forward only')
        ws = eng._cur_ws
        T, d = ws['T'], eng.d
        p0 = eng.params[0]
        alias = p0.grad is not None and p0.grad.data_ptr() == eng.grad_views[0].data_ptr()
        eng._select_grads(alias)
        conv = lambda g: None if g is None else g.reshape(T, d).to(eng.xdt).contiguous()
        eng.zero_accumulated_grads()
        if ctx.has_dec:
            eng.backward(conv(d_dec) if d_dec is not None else torch.zeros(T, d, dtype=eng.xdt, device=eng.device), conv(d_enc))
        else:
            eng.backward(None, conv(d_enc))
        grads = eng.grad_views_of(eng.Gcur)[:ctx.nparams]
        eng._select_grads(False)
        d_embeds = eng._alt_grad.float().view(ctx.alt_shape) if ctx.alt_shape is not None else None
        return (None,) * 7 + (d_embeds,) + tuple(grads)
