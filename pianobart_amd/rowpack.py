"""Dead-row compaction of a pre-train batch for the fused step (Engine.loss_and_grads): the host side of csrc/pb_rowmap.hip.

A row of the padded batch is dead when nothing reads what is computed for it (an encoder row that is masked as a key; a decoder
row that is masked as a key and carries no loss term); DESIGN.md 5 "Packed rows" has the argument why dropping them changes no
result. `pack_batch` asks the device for the per-sequence counts, plans the packed layout on the host (one small device -> host
copy and a stream synchronisation per step: the packed row counts size every later launch) and gathers the packed inputs."""
import os

import numpy as np
import torch

from . import ops
from ._lib import PB_BF16

PACK_TILE = 256                 # packed row counts are rounded up to whole GEMM tiles
PACK_MIN_GAIN = 0.97            # stay dense unless at least 3 % of the rows go
SUB_LAST = int(os.environ.get('PB_SUB_LAST', '1'))     # last decoder layer: query side and LM heads on the loss rows only
SUB_MIN_GAIN = 0.75             # ... unless more than 3/4 of the decoder rows carry a loss term
X3_PACK = int(os.environ.get('PB_X3_PACK', '1'))            # dead-row compaction also for the bf16x3 instantiation (0: its step stays padded, for A/B)
ORDER_PAIRS = int(os.environ.get('PB_ORDER_PAIRS', '1'))   # attention grids take the (batch, head) pairs longest first (0: batch order, for A/B)
ORDER_CAUSAL = int(os.environ.get('PB_ORDER_CAUSAL', '1'))  # ... the decoder's causal self-attention too (0: batch order there, the choice before round 6's rotated row blocks)


def plan_packed_rows(live, S, tile=PACK_TILE):
    """live[b] = rows of sequence b that must be kept (<= S). Returns (Tp, off, length): the packed side has Tp rows, a multiple of
    `tile` (the GEMM row tile; len(live) * S must be one) and no more than len(live) * S; sequence b owns rows off[b] .. off[b] +
    length[b] - 1, length[b] >= live[b]: the Tp - sum(live) rows of slack are handed out as dead rows of the sequences that have
    some (first come, first served), so that no sequence grows beyond S."""
    live = np.asarray(live, dtype=np.int64)
    cap = np.broadcast_to(np.asarray(S, dtype=np.int64), live.shape)     # S may also be one capacity per sequence
    total = int(live.sum())
    Tp = min(-(-max(total, 1) // tile) * tile, int(cap.sum()))
    cum = np.minimum(np.cumsum(cap - live), Tp - total)
    length = live + np.diff(np.concatenate([[0], cum]))
    return Tp, np.concatenate([[0], np.cumsum(length)[:-1]]), length


class RowPack:
    """One packed batch: B, S, Te / Td (rows kept on the encoder / decoder side), the packed inputs enc16 / dec16 / tgt16 /
    loss_mask, src_* (row b*S + s of every packed row in the padded batch), inv_* (packed row of every (b, s), or -1) and the PackedRows
    descriptors of the encoder self-, decoder self- and cross-attention."""



def dispatch_order(cost, H):
    """Order in which an attention grid should take the (batch, head) pairs of a packed batch (ops.PackedRows.order): pairs sorted by cost,
    longest first (all heads of a sequence cost the same), dealt eight at a time in snake order so that the 8 XCDs -- the grid gives
    slot 8 g + x to XCD x -- receive equal sums. cost: (B,) array-like. Returns int32 (B * H,), entry = b * H + h."""
    cost = np.asarray(cost, dtype=np.float64)
    pairs = (np.argsort(-cost, kind='stable')[:, None] * H + np.arange(H)[None, :]).reshape(-1)
    n = pairs.size
    if n % 8 == 0:
        g = pairs.reshape(-1, 8).copy()
        g[1::2] = g[1::2, ::-1]
        pairs = g.reshape(-1)
    return pairs.astype(np.int32)


_TICKET = [0]


def _packable(eng):
    """The instantiations whose attention kernels take packed rows: bf16 (pipelined family, head_dim 64 / 96 / 128) and, since round 6, bf16x3 (fused
    split-bf16 kernels, head_dim 32 / 64 / 128)."""
    if not eng.use_flash:
        return False
    if getattr(eng, 'x3', False):
        return X3_PACK and eng.hd in (32, 64, 128)
    return eng.code == PB_BF16 and eng.hd in (64, 96, 128)


def _issue_ticket(loss_mask, emask, dmask):
    """A prefetch request and the batch it was made for are matched by a ticket, not by addresses: the kernels write the masks through raw
    pointers (no version bump) and the caching allocator hands the next batch the same addresses. The ticket queues up on the
    three tensor OBJECTS; the step that consumes the batch takes the front one off."""
    _TICKET[0] += 1
    for t in (loss_mask, emask, dmask):
        q = getattr(t, '_pb_tickets', None)
        if q is None:
            q = t._pb_tickets = []
        q.append(_TICKET[0])
    return _TICKET[0]


def _take_ticket(loss_mask, emask, dmask):
    """The front ticket the three tensors share (popped), or None if they carry none / disagree."""
    qs = [getattr(t, '_pb_tickets', None) for t in (loss_mask, emask, dmask)]
    if any(not q for q in qs) or len({q[0] for q in qs}) != 1:
        for q in qs:
            if q:
                q.clear()
        return None
    return [q.pop(0) for q in qs][0]


def prefetch_counts(eng, loss_mask, emask, dmask, stream=None):
    """Ask for the per-sequence row counts of a batch the caller will hand to Engine.loss_and_grads NEXT, so that pack_batch finds them on
    the host instead of draining the step's stream for them (one kernel over the masks + a 1 KiB copy). `stream`: a stream on which the
    three mask tensors are ready (e.g. the one that produced them); default = a private stream that waits for everything the current
    stream has been given so far. The tensors must not be modified between this call and the step; request and batch are matched by a
    ticket that travels on the tensor objects (a batch without the right ticket just takes the synchronous path)."""
    B, S = emask.shape[:2]
    if not (_packable(eng) and (B * S) % PACK_TILE == 0 and eng.mlm is not None):
        return
    pf = eng._pack_pf_state
    if pf is None or pf['shape'] != (B, S):
        pf = eng._pack_pf_state = dict(shape=(B, S), stream=torch.cuda.Stream(device=eng.device), turn=0,
                                       counts=[torch.empty(B, 8, dtype=torch.int32, device=eng.device) for _ in range(2)],
                                       counts_h=[torch.empty(B, 8, dtype=torch.int32).pin_memory() for _ in range(2)])
        eng._pack_prefetch.clear()
    if len(eng._pack_prefetch) >= 2:                     # two requests in flight at most (two buffers): the older one is dropped
        eng._pack_prefetch.pop(0)
    k = pf['turn'] = pf['turn'] ^ 1
    st = stream
    if st is None:
        st = pf['stream']
        st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        ops.rowmap_count(emask, dmask, loss_mask.reshape(B, S, 8), pf['counts'][k])
        pf['counts_h'][k].copy_(pf['counts'][k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(st)
    eng._pack_prefetch.append(dict(ticket=_issue_ticket(loss_mask, emask, dmask), event=ev, counts_h=pf['counts_h'][k]))


def pack_batch(eng, enc16, dec16, tgt16, loss_mask, emask, dmask):
    """Dead-row compaction of one batch (csrc/pb_rowmap.hip has the argument why it changes no result): returns a RowPack with
    the packed inputs and the row descriptors of the three attention forms, or None when the step must stay dense (unsupported
    shape, a decoder mask that is not a prefix mask, or nothing to gain). Costs one small device -> host copy and a stream
    synchronisation: the packed row counts size every later launch."""
    B, S = enc16.shape[:2]
    T = B * S
    if not (_packable(eng) and T % PACK_TILE == 0 and emask is not None
            and dmask is not None and eng.mlm is not None):
        return None
    st = eng._pack_state
    if st is None or st['key'] != (B, S):
        dev = eng.device
        i32 = lambda *shape: torch.empty(*shape, dtype=torch.int32, device=dev)
        st = eng._pack_state = dict(key=(B, S), counts=i32(B, 8), counts_h=torch.empty(B, 8, dtype=torch.int32).pin_memory(),
                                     desc=i32(8, B), desc_h=[torch.empty(8, B, dtype=torch.int32).pin_memory() for _ in range(4)], desc_turn=0,
                                     order=i32(4, B * eng.H), order_h=[torch.empty(4, B * eng.H, dtype=torch.int32).pin_memory() for _ in range(4)],
                                     src_e=i32(T), pos_e=i32(T), inv_e=i32(T), src_d=i32(T), pos_d=i32(T), inv_d=i32(T), src_s=i32(T), idx_s=i32(T),
                                     tgt16_s=torch.empty(T, 8, dtype=torch.int16, device=dev), lm_s=torch.empty(T, 8, dtype=torch.float32, device=dev),
                                     enc16=torch.empty(T, 8, dtype=torch.int16, device=dev), dec16=torch.empty(T, 8, dtype=torch.int16, device=dev),
                                     tgt16=torch.empty(T, 8, dtype=torch.int16, device=dev), lm=torch.empty(T, 8, dtype=torch.float32, device=dev))
    lm3 = loss_mask.reshape(B, S, 8)
    ticket = _take_ticket(loss_mask, emask, dmask)
    pf = None
    while eng._pack_prefetch and ticket is not None:
        cand = eng._pack_prefetch.pop(0)
        if cand['ticket'] == ticket:
            pf = cand
            break                                        # older requests (batches that never reached a step) are dropped on the way
    if ticket is None:
        eng._pack_prefetch.clear()                       # a batch nobody announced: whatever is queued belongs to batches that will not come
    if pf is not None:
        # the per-sequence counts of this batch were requested ahead of time (prefetch_counts): the host has them, or waits for that one
        # small copy only -- the step's stream is never drained, so the launches of this step queue up behind the previous one
        pf['event'].synchronize()
        c = pf['counts_h'].numpy().astype(np.int64)
        eng._await_updates(0)
        eng.refresh_shadow()
        eng.build_ptab()
        eng._tables_ready = True
    else:
        ops.rowmap_count(emask, dmask, lm3, st['counts'])
        st['counts_h'].copy_(st['counts'], non_blocking=True)
        # work of the step that does not depend on the row counts goes in front of the wait: the GPU projects the Octuple table while the
        # host plans the packing
        eng._await_updates(0)
        eng.refresh_shadow()
        eng.build_ptab()
        eng._tables_ready = True
        torch.cuda.current_stream().synchronize()
        c = st['counts_h'].numpy().astype(np.int64)
    if not c[:, 3].all():
        return None
    if int(c[:, 4].sum()) == 0:
        # no loss position in the whole batch: pretrain.py:116-117 divides 0 by 0 in every head, the loss and EVERY gradient are NaN. The
        # packed step would run its last layer on zero rows and leave the gradients at 0: such a batch takes the padded step, which mirrors
        # the reference
        return None
    (Te, off_e, len_e), (Td, off_d, len_d) = plan_packed_rows(c[:, 0], S), plan_packed_rows(c[:, 2], S)
    if Te + Td > PACK_MIN_GAIN * 2 * T:
        return None
    # the last decoder layer's query side (cross-attention, FFN) and the LM heads are only needed on rows that carry a loss term:
    # every other layer's output is a later layer's key / value input, the last layer's is read by the loss alone
    Ts, off_s, len_s = plan_packed_rows(c[:, 4], len_d)
    sub = SUB_LAST and eng.ND > 0 and Ts <= SUB_MIN_GAIN * Td
    if not sub:
        off_s, len_s = off_d, len_d
    # the host may run a step ahead of the device (prefetch_counts): the pinned staging of the offsets is a ring, so that this step's
    # values are not overwritten before their stream-ordered copy has run (a slot comes round again after three more steps, and the
    # host never gets further than two steps ahead: a prefetched count waits for the step before the previous one)
    st['desc_turn'] = (st['desc_turn'] + 1) % len(st['desc_h'])
    desc_h = st['desc_h'][st['desc_turn']]
    desc_h.copy_(torch.from_numpy(np.stack([off_e, len_e, c[:, 0], off_d, len_d, c[:, 1], off_s, len_s]).astype(np.int32)))
    desc = st['desc']
    desc.copy_(desc_h, non_blocking=True)
    ops.rowmap_build(emask, None, desc[0], desc[1], st['src_e'], st['pos_e'], st['inv_e'])
    ops.rowmap_build(dmask, lm3, desc[3], desc[4], st['src_d'], st['pos_d'], st['inv_d'])
    pk = RowPack()
    pk.B, pk.S, pk.Te, pk.Td = B, S, Te, Td
    pk.src_e, pk.src_d, pk.inv_e, pk.inv_d = st['src_e'], st['src_d'], st['inv_e'], st['inv_d']
    pk.enc16, pk.dec16, pk.tgt16, pk.loss_mask = st['enc16'][:Te], st['dec16'][:Td], st['tgt16'][:Td], st['lm'][:Td]
    ops.gather_rows16(enc16, st['src_e'], pk.enc16, Te, 16)
    ops.gather_rows16(dec16, st['src_d'], pk.dec16, Td, 16)
    ops.gather_rows16(tgt16, st['src_d'], pk.tgt16, Td, 16)
    ops.gather_rows16(lm3, st['src_d'], pk.loss_mask, Td, 32)
    me, md = int(len_e.max()), int(len_d.max())
    vis_e, vis_d = c[:, 0], c[:, 1]
    pk.pairs = (int((len_e * vis_e).sum()), int((vis_d * vis_d // 2 + (len_d - vis_d) * vis_d).sum()), int((len_d * vis_e).sum()))
    # dispatch orders of the (batch, head) pairs, longest first (the costs spread 4x over a batch; a static grid in batch order ends on
    # whatever lies last): encoder self-, decoder self-, cross-attention and the last layer's cross-attention on the loss rows
    order = st['order']
    if ORDER_PAIRS:
        order_h = st['order_h'][st['desc_turn']]
        costs = (len_e * vis_e, vis_d * vis_d // 2 + (len_d - vis_d) * vis_d, len_d * vis_e, (len_s if sub else len_d) * vis_e)
        order_h.copy_(torch.from_numpy(np.stack([dispatch_order(x, eng.H) for x in costs])))
        order.copy_(order_h, non_blocking=True)
    o = (lambda i: order[i]) if ORDER_PAIRS else (lambda i: None)
    pk.enc = ops.PackedRows(desc[0], desc[1], desc[0], desc[1], desc[2], me, me, 'enc', order=o(0))
    # causal: until round 6 batch order (longest first measured 4 - 10 % SLOWER: with the row blocks of a head in order one shader engine of an XCD collected the long ones,
    # pb_fa_tiles.h block_map); with the rotated row blocks longest first wins here too (forward 94.5 -> 88 us, backward pair 252 -> 239: profiles/r06_attention_dispatch_trace.txt)
    pk.dec = ops.PackedRows(desc[3], desc[4], desc[3], desc[4], desc[5], md, md, 'dec', order=o(1) if ORDER_CAUSAL else None)
    pk.cross = ops.PackedRows(desc[3], desc[4], desc[0], desc[1], desc[2], md, me, 'cross', order=o(2))
    pk.sub = None
    if sub:
        sb = pk.sub = RowPack()
        sb.T, sb.src, sb.idx = Ts, st['src_s'], st['idx_s']
        ops.rowmap_build_sub(lm3, st['inv_d'], desc[6], desc[7], sb.src, sb.idx)
        sb.tgt16, sb.loss_mask = st['tgt16_s'][:Ts], st['lm_s'][:Ts]
        ops.gather_rows16(pk.tgt16, sb.idx, sb.tgt16, Ts, 16)
        ops.gather_rows16(pk.loss_mask, sb.idx, sb.loss_mask, Ts, 32)
        ms = int(len_s.max())
        sb.cross = ops.PackedRows(desc[6], desc[7], desc[0], desc[1], desc[2], ms, me, 'cross', order=o(3))
        pk.pairs = pk.pairs[:2] + (pk.pairs[2] * (eng.ND - 1) // eng.ND + int((len_s * vis_e).sum()) // eng.ND,)     # layer mean
    return pk

