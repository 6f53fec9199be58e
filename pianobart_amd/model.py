"""Drop-in PianoBART classes for MI355X.

Same public surface as the reference (PianoBart.py:9-91, model.py:14-126): `Embeddings`, `PianoBart`,
`MLM`, `PianoBartLM`, `sampling`, `nucleus`, and the same `state_dict` layout. The
modules below only *hold* the parameters under the reference's names; all arithmetic runs in the
hand-written HIP kernels of libpianobart_hip.so through `pianobart_amd.engine.Engine`. There is no
CPU execution path: calling forward with CPU tensors raises.
"""
import math
import random
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import LIB, PBError

CLASSES = ['Bar', 'Position', 'Instrument', 'Pitch', 'Duration', 'Velocity', 'TimeSig', 'Tempo']


def checkpoint_state_dict(sd, module=None):
    """The `state_dict` of a reference checkpoint, ready for `load_state_dict`.

    Under `nn.DataParallel` the reference saves `self.model.state_dict()` of the WRAPPER (finetune_generation.py:276-285,
    finetune.py save_checkpoint), so every key carries a leading `module.`; its own `demo.py:128-129` then loads that file with
    `strict=False` and silently keeps the random initialisation. Here the prefix is stripped (with a message), and -- when `module`
    is given -- a file none of whose keys name a parameter of `module` is reported instead of being "loaded"."""
    if sd and all(k.startswith('module.') for k in sd):
        print("   [pianobart_amd] checkpoint was saved from an nn.DataParallel wrapper: stripping the 'module.' prefix of its %d keys" % len(sd))
        sd = type(sd)((k[len('module.'):], v) for k, v in sd.items())
    if module is not None:
        own = set(module.state_dict().keys())
        hit = sum(k in own for k in sd)
        if hit == 0:
            print("   [pianobart_amd] WARNING: none of the checkpoint's %d keys names a parameter of %s (first key: %r): nothing will be "
                  "loaded under strict=False" % (len(sd), type(module).__name__, next(iter(sd), None)))
    return sd


class BartConfig:
    """Stand-in for transformers.BartConfig with the attributes the reference reads (main.py:39-47 sets the
    first eight; the rest are BartConfig defaults). Any object with these attribute names is accepted."""

    def __init__(self, max_position_embeddings=1024, d_model=1024, encoder_layers=12, encoder_ffn_dim=4096,
                 encoder_attention_heads=16, decoder_layers=12, decoder_ffn_dim=4096, decoder_attention_heads=16,
                 vocab_size=50265, dropout=0.1, attention_dropout=0.0, activation_dropout=0.0,
                 activation_function="gelu", init_std=0.02, scale_embedding=False, pad_token_id=1, **kw):
        self.max_position_embeddings = max_position_embeddings
        self.d_model = d_model
        self.encoder_layers = encoder_layers
        self.encoder_ffn_dim = encoder_ffn_dim
        self.encoder_attention_heads = encoder_attention_heads
        self.decoder_layers = decoder_layers
        self.decoder_ffn_dim = decoder_ffn_dim
        self.decoder_attention_heads = decoder_attention_heads
        self.vocab_size = vocab_size
        self.dropout = dropout
        self.attention_dropout = attention_dropout
        self.activation_dropout = activation_dropout
        self.activation_function = activation_function
        self.init_std = init_std
        self.scale_embedding = scale_embedding
        self.pad_token_id = pad_token_id
        for k, v in kw.items():
            setattr(self, k, v)


# ---- parameter containers named exactly like transformers' BartModel (modeling_bart.py) -------------
class _Attention(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj = nn.Linear(d, d)
        self.v_proj = nn.Linear(d, d)
        self.q_proj = nn.Linear(d, d)
        self.out_proj = nn.Linear(d, d)


class _EncLayer(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.self_attn = _Attention(d)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, f)
        self.fc2 = nn.Linear(f, d)
        self.final_layer_norm = nn.LayerNorm(d)


class _DecLayer(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.self_attn = _Attention(d)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.encoder_attn = _Attention(d)
        self.encoder_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, f)
        self.fc2 = nn.Linear(f, d)
        self.final_layer_norm = nn.LayerNorm(d)


class _Stack(nn.Module):
    def __init__(self, cfg, shared, decoder):
        super().__init__()
        d = cfg.d_model
        self.embed_tokens = shared      # dead 50265 x d table kept for checkpoint compatibility (never read)
        self.embed_positions = nn.Embedding(cfg.max_position_embeddings + 2, d)
        if decoder:
            self.layers = nn.ModuleList([_DecLayer(d, cfg.decoder_ffn_dim) for _ in range(cfg.decoder_layers)])
        else:
            self.layers = nn.ModuleList([_EncLayer(d, cfg.encoder_ffn_dim) for _ in range(cfg.encoder_layers)])
        self.layernorm_embedding = nn.LayerNorm(d)


class _BartParams(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.shared = nn.Embedding(cfg.vocab_size, cfg.d_model, padding_idx=cfg.pad_token_id)
        self.encoder = _Stack(cfg, self.shared, decoder=False)
        self.decoder = _Stack(cfg, self.shared, decoder=True)
        std = cfg.init_std
        for m in self.modules():            # BartPreTrainedModel._init_weights distributions
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(0.0, std)
                m.bias.data.zero_()
            elif isinstance(m, nn.Embedding):
                m.weight.data.normal_(0.0, std)
                if m.padding_idx is not None:
                    m.weight.data[m.padding_idx].zero_()


class Embeddings(nn.Module):
    """PianoBart.py:9-16 (parameter holder; lut(x)*sqrt(d_model) is folded into the projected table)."""

    def __init__(self, n_token, d_model):
        super().__init__()
        self.lut = nn.Embedding(n_token, d_model)
        self.d_model = d_model

    def forward(self, x):
        """lut(x) * sqrt(d_model) (PianoBart.py:15-16) for direct callers: a HIP row gather + scale, differentiable in `lut`. Inside
        PianoBart.forward the eight embeddings never run on their own: they are folded into the projected Octuple table."""
        from . import heads
        e = heads.gather_rows(self.lut.weight, x)
        return heads.mul(e, torch.full_like(e, math.sqrt(self.d_model)))


def _check_cfg(cfg):
    d = cfg.d_model
    if d % 4 != 0 or d > 2048:
        raise PBError('d_model=%d unsupported by the HIP row kernels (need a multiple of 4, <= 2048)' % d)
    if cfg.encoder_attention_heads != cfg.decoder_attention_heads or d % cfg.encoder_attention_heads != 0:
        raise PBError('encoder/decoder head counts must match and divide d_model')
    if (d // cfg.encoder_attention_heads) % 8 != 0:
        raise PBError('head_dim must be a multiple of 8')
    if getattr(cfg, 'activation_function', 'gelu') != 'gelu':
        raise PBError('only the exact-erf "gelu" activation of the reference is implemented')
    if getattr(cfg, 'attention_dropout', 0.0) != 0.0 or getattr(cfg, 'activation_dropout', 0.0) != 0.0:
        raise PBError('attention_dropout / activation_dropout must be 0 (reference defaults)')
    if getattr(cfg, 'scale_embedding', False):
        raise PBError('scale_embedding=True is not used by the reference and not implemented')


class PianoBart(nn.Module):
    """PianoBart.py:19-91. `precision`: "bf16" (throughput, bf16 MFMA), "fp32" (exact-f32 parity path) or "bf16x3" (f32 storage and row
    kernels, every GEMM as split-bf16 triples on the bf16 matrix cores: parity-grade at a multiple of the f32-MFMA rate)."""

    def __init__(self, bartConfig, e2w, w2e, precision='bf16'):
        super().__init__()
        _check_cfg(bartConfig)
        self.bart = _BartParams(bartConfig)
        self.hidden_size = bartConfig.d_model
        self.bartConfig = bartConfig
        self.n_tokens = []
        self.classes = list(CLASSES)
        for key in self.classes:
            self.n_tokens.append(len(e2w[key]))
        if self.n_tokens != ops.SEG_SIZES:
            raise PBError('vocabulary sizes %s differ from the Octuple layout compiled into the kernels' % self.n_tokens)
        self.emb_sizes = [256] * 8
        self.e2w = e2w
        self.w2e = w2e
        self.bar_pad_word = self.e2w['Bar']['Bar <PAD>']
        mk = lambda tag: np.array([self.e2w[e]['%s <%s>' % (e, tag)] for e in self.classes], dtype=np.int64)
        self.mask_word_np = mk('MASK')
        self.pad_word_np = mk('PAD')
        self.sos_word_np = mk('SOS')
        self.eos_word_np = mk('EOS')
        self.word_emb = nn.ModuleList([Embeddings(self.n_tokens[i], self.emb_sizes[i]) for i in range(8)])
        self.encoder_linear = nn.Linear(int(np.sum(self.emb_sizes)), bartConfig.d_model)
        self.decoder_linear = self.encoder_linear
        self.decoder_emb = None
        self.precision = precision
        object.__setattr__(self, '_engine', None)
        object.__setattr__(self, '_engine_owner', None)

    # -- engine plumbing ---------------------------------------------------------------------
    def _get_engine(self):
        if self._engine_owner is not None:
            return self._engine_owner._get_engine()
        if self._engine is None:
            from .engine import Engine
            object.__setattr__(self, '_engine', Engine(self, None, self.precision))
        return self._engine

    def forward(self, input_ids_encoder, input_ids_decoder=None, encoder_attention_mask=None,
                decoder_attention_mask=None, output_hidden_states=True, generate=False):
        eng = self._get_engine()
        if self.decoder_emb is not None and input_ids_decoder is not None:
            # PianoBart.py:65-66,71: decoder_linear(decoder_emb(labels)) with Embeddings = lut(x) * sqrt(d_model) -- computed as a
            # row gather from the projected label table sqrt(d_model) * lut @ W^T (n_labels x d), differentiable in lut, W and b
            from . import heads
            table = heads.linear(self.decoder_emb.lut.weight, self.decoder_linear.weight, None, alpha=math.sqrt(self.decoder_emb.d_model))
            e = heads.gather_rows(table, input_ids_decoder, self.decoder_linear.bias)
            dec_h, enc_h = eng.module_forward_hidden(input_ids_encoder, None, encoder_attention_mask, decoder_attention_mask, self.training,
                                                     dec_embeds=e)
            return SimpleNamespace(last_hidden_state=dec_h, encoder_last_hidden_state=enc_h)
        dec_h, enc_h = eng.module_forward_hidden(input_ids_encoder, input_ids_decoder, encoder_attention_mask,
                                                 decoder_attention_mask, self.training)
        if input_ids_decoder is None:
            return SimpleNamespace(last_hidden_state=enc_h)
        return SimpleNamespace(last_hidden_state=dec_h, encoder_last_hidden_state=enc_h)

    def get_rand_tok(self):
        rand = [0] * 8
        for i in range(8):
            rand[i] = random.choice(range(self.n_tokens[i]))
        return np.array(rand)

    def change_decoder_embedding(self, new_embedding, new_linear=None):
        self.decoder_emb = new_embedding
        if new_linear is not None:
            self.decoder_linear = new_linear


class MLM(nn.Module):
    """model.py:109-126 (parameter holder; the 8 heads run as one fused d x 1280 GEMM)."""

    def __init__(self, e2w, n_tokens, hidden_size):
        super().__init__()
        self.proj = nn.ModuleList([nn.Linear(hidden_size, n_tokens[i]) for i, _ in enumerate(e2w)])
        self.e2w = e2w

    def forward(self, y):
        """model.py:119-126 for direct callers: y = pianobart(...) output (anything with .last_hidden_state) or a (B,S,d) tensor ->
        list of 8 (B,S,n_i) f32 logits through the exact-f32 HIP GEMM. PianoBartLM.forward runs the 8 heads as ONE fused GEMM instead."""
        from . import heads
        h = y.last_hidden_state if hasattr(y, 'last_hidden_state') else y
        return [heads.linear(h, self.proj[i].weight, self.proj[i].bias) for i, _ in enumerate(self.e2w)]


# -- nucleus sampling: host-side numpy exactly like the reference, so the RNG stream matches ----------
def nucleus(probs, p):
    """model.py:84-98 (mutates probs in place; draws from the global np.random)."""
    probs /= (sum(probs) + 1e-5)
    sorted_probs = np.sort(probs)[::-1]
    sorted_index = np.argsort(probs)[::-1]
    cusum_sorted_probs = np.cumsum(sorted_probs)
    after_threshold = cusum_sorted_probs > p
    if sum(after_threshold) > 0:
        last_index = np.where(after_threshold)[0][0] + 1
        candi_index = sorted_index[:last_index]
    else:
        candi_index = sorted_index[0:1]
    candi_probs = [probs[i] for i in candi_index]
    candi_probs /= sum(candi_probs)
    word = np.random.choice(candi_index, size=1, p=candi_probs)[0]
    return word


_SAMPLE_TAB = {}


def _sample_tables():
    """Constants and scratch of PianoBartLM.sample_row: per-element temperatures, the (8, 272) probability rows, the native call's arrays."""
    if not _SAMPLE_TAB:
        n = [ops.SEG_OFF[j + 1] - ops.SEG_OFF[j] for j in range(8)]
        width = (max(n) + 15) // 16 * 16
        n_a, p_a = np.asarray(n, dtype=np.int32), np.asarray(PianoBartLM.SAMPLE_P, dtype=np.float32)
        _SAMPLE_TAB.update(n_a=n_a, p_a=p_a, n_p=n_a.ctypes.data, p_p=p_a.ctypes.data, out=np.zeros(8, dtype=np.int32), tie=np.zeros(1, dtype=np.int32))
        _SAMPLE_TAB.update(n=n, probs=torch.zeros(8, width, dtype=torch.float32),
                           tvec=torch.cat([torch.full((n[j],), float(PianoBartLM.SAMPLE_T[j]), dtype=torch.float32) for j in range(8)]))
    return _SAMPLE_TAB


def _nucleus_with_draw(probs, p, u):
    """_nucleus_fast with the uniform draw handed in (PianoBartLM.sample_row draws the 8 of a position at once)."""
    probs = probs / (np.cumsum(probs)[-1] + 1e-5)
    order = np.argsort(probs)[::-1]
    after = np.cumsum(probs[order]) > p
    cand = order[:int(np.argmax(after)) + 1] if after.any() else order[0:1]
    q = probs[cand]
    q = q / np.cumsum(q)[-1]
    cdf = q.astype(np.float64).cumsum()
    cdf /= cdf[-1]
    return int(cand[int(cdf.searchsorted(u, side='right'))])


def _nucleus_fast(probs, p):
    """The same draw as nucleus() for the same probs / global np.random state, without its Python-level loops: builtin sum() over a
    float32 array is a left-to-right float32 accumulation = np.cumsum(..)[-1]; np.random.choice(c, size=1, p=q) is, in RandomState,
    cdf = q.astype(f64).cumsum(); cdf /= cdf[-1]; c[cdf.searchsorted(random_sample(1), 'right')]. Checked draw for draw (values and
    RNG stream) against nucleus() in tests/test_model_cpu.py; 0.30 -> 0.1 ms of host time per generated position."""
    probs /= (np.cumsum(probs)[-1] + 1e-5)
    order = np.argsort(probs)[::-1]
    after = np.cumsum(probs[order]) > p
    cand = order[:int(np.argmax(after)) + 1] if after.any() else order[0:1]
    q = probs[cand]
    q = q / np.cumsum(q)[-1]
    cdf = q.astype(np.float64).cumsum()
    cdf /= cdf[-1]
    return cand[int(cdf.searchsorted(np.random.random_sample(1), side='right')[0])]


def sampling(logit, p=None, t=1.0):
    """model.py:101-107."""
    logit = logit.squeeze()
    probs = torch.softmax(logit / t, dim=-1)
    probs = probs.cpu().detach().numpy()
    return nucleus(probs, p=p)


class PianoBartLM(nn.Module):
    """model.py:14-78. Train branch returns a mutable list of 8 (B,S,n_i) f32 tensors with autograd;
    generate=True runs the KV-cached HIP decode (same tokens as the reference's full re-run)."""

    def __init__(self, pianobart: PianoBart):
        super().__init__()
        self.pianobart = pianobart
        self.mask_lm = MLM(self.pianobart.e2w, self.pianobart.n_tokens, self.pianobart.hidden_size)
        object.__setattr__(self, '_engine', None)
        # plain attribute (bypass nn.Module registration: the LM must not become a child of its child)
        object.__setattr__(pianobart, '_engine_owner', self)
        object.__setattr__(pianobart, '_engine', None)

    def _get_engine(self):
        if self._engine is None:
            from .engine import Engine
            object.__setattr__(self, '_engine', Engine(self.pianobart, self.mask_lm, self.pianobart.precision))
        return self._engine

    def forward(self, input_ids_encoder, input_ids_decoder=None, encoder_attention_mask=None,
                decoder_attention_mask=None, generate=False, device_num=-1):
        eng = self._get_engine()
        if not generate:
            logits = eng.module_forward_logits(input_ids_encoder, input_ids_decoder, encoder_attention_mask,
                                               decoder_attention_mask, self.training)
            return [logits[..., ops.SEG_OFF[i]:ops.SEG_OFF[i + 1]] for i in range(8)]
        if input_ids_encoder.shape[0] != 1:
            print("ERROR")
            exit(-1)
        out = eng.generate(input_ids_encoder, encoder_attention_mask, self.sample_row, sampler=dict(T=self.SAMPLE_T, P=self.SAMPLE_P))
        # model.py:33-36: the result lives on `cuda:device_num`, or on the CPU for device_num == -1
        return out.cpu() if device_num == -1 else out.to(torch.device('cuda', device_num))

    # model.py:68-78 -- temperatures / nucleus thresholds per head
    SAMPLE_T = [1.2, 1.2, 5, 1, 2, 5, 5, 1.2]
    SAMPLE_P = [1, 1, 1, 0.9, 0.9, 1, 1, 0.9]

    def sample_row(self, row_logits):
        """row_logits: (1280,) f32 CPU tensor of one position; returns the 8 sampled ids (model.py:68-78). sampling()'s own tensor ops
        on the host row -- the division by the temperature (one call with a per-element temperature vector: the same quotients) and a
        1-D softmax per head -- then nucleus() for all 8 heads in one native call (pb_nucleus_rows: numpy's arithmetic order and
        precision; ties among candidates go back to the numpy code), fed the 8 uniform draws np.random.choice would have made. Checked
        draw for draw and RNG state for RNG state against sampling() in tests/test_model_cpu.py. 0.31 -> 0.1 ms of host time per
        generated position, which sits in series with the GPU's ~0.3 ms."""
        tab = _sample_tables()
        y = row_logits / tab['tvec']
        probs = tab['probs']
        for j in range(8):                                           # 1-D calls: a 2-D softmax would open an OpenMP region per position
            torch.softmax(y[ops.SEG_OFF[j]:ops.SEG_OFF[j + 1]], dim=-1, out=probs[j, :tab['n'][j]])
        # the 8 draws np.random.choice would make, in head order (RandomState fills a request sequentially: the same stream as 8 calls)
        u = np.random.random_sample(8)
        out, tie = tab['out'], tab['tie']
        LIB.call('pb_nucleus_rows', probs.data_ptr(), probs.shape[1], tab['n_p'], tab['p_p'], u.ctypes.data, 8, out.ctypes.data, tie.ctypes.data)
        if tie[0]:                                                   # equal probabilities among a head's candidates: numpy's own order decides
            pn = probs.numpy()
            for j in range(8):
                if tie[0] >> j & 1:
                    out[j] = _nucleus_with_draw(pn[j, :tab['n'][j]], self.SAMPLE_P[j], u[j])
        return torch.from_numpy(out.astype(np.int64))

    def sample(self, x, index):
        t, p = self.SAMPLE_T, self.SAMPLE_P
        return torch.tensor([sampling(x[j][:, index, :], p[j], t[j]) for j in range(8)])


# ---------------------------------------------------------------------------------------------------------------------
# Fine-tune heads. Same constructor arguments, attribute names and state_dict keys as the reference
# (model.py:128-272); the nn.Linear / nn.Sequential members are parameter holders, the arithmetic runs through the HIP ops of
# heads.py (exact f32 on top of the backbone's hidden states).
class SelfAttention(nn.Module):
    """model.py:128-143: r-aspect attention pooling weights, softmax over the sequence axis."""

    def __init__(self, input_dim, da, r):
        super().__init__()
        self.ws1 = nn.Linear(input_dim, da, bias=False)
        self.ws2 = nn.Linear(da, r, bias=False)

    def scores(self, h):
        """softmax(ws2(tanh(ws1(h))), dim=1): (B, S, r), i.e. attn_mat before the reference's permute(0, 2, 1)."""
        from . import heads
        return heads.softmax_dim1(heads.linear(heads.act(heads.linear(h, self.ws1.weight), 'tanh'), self.ws2.weight))

    def forward(self, h):
        return self.scores(h).permute(0, 2, 1)


class SequenceClassification(nn.Module):
    """model.py:165-218: the backbone is run with decoder input = encoder input (model.py:203), attention-pooled into r = 4
    aspects, flattened and classified by Dropout(0.1) -> Linear(hs r, 256) -> ReLU -> Linear(256, class_num)."""

    def __init__(self, pianobart, class_num, hs, da=128, r=4):
        super().__init__()
        self.pianobart = pianobart
        self.attention = SelfAttention(hs, da, r)
        self.classifier = nn.Sequential(nn.Dropout(0.1), nn.Linear(hs * r, 256), nn.ReLU(), nn.Linear(256, class_num))

    def forward(self, input_ids_encoder, encoder_attention_mask=None):
        from . import heads
        x = self.pianobart(input_ids_encoder=input_ids_encoder, input_ids_decoder=input_ids_encoder,
                           encoder_attention_mask=encoder_attention_mask, decoder_attention_mask=encoder_attention_mask).last_hidden_state
        m = heads.pool(self.attention.scores(x), x)                     # torch.bmm(attn_mat, x): (B, r, hs)
        flat = m.reshape(m.shape[0], -1)
        c = self.classifier
        y = heads.dropout(flat, c[0].p, self.training)
        y = heads.act(heads.linear(y, c[1].weight, c[1].bias), 'relu')
        return heads.linear(y, c[3].weight, c[3].bias)


class Excitation(nn.Module):
    """model.py:220-232 (squeeze-excitation gate; commented out of both classifiers in the reference, kept for API parity)."""

    def __init__(self, channel_dim, reduction=16):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(channel_dim, channel_dim // reduction), nn.ReLU(), nn.Linear(channel_dim // reduction, channel_dim), nn.Sigmoid())

    def forward(self, x):
        from . import heads
        y = heads.act(heads.linear(x, self.fc[0].weight, self.fc[0].bias), 'relu')
        y = heads.act(heads.linear(y, self.fc[2].weight, self.fc[2].bias), 'sigmoid')
        return heads.mul(x, y)


class TokenClassification(nn.Module):
    """model.py:236-272: per-token Dropout(0.1) -> Linear(hs, 256) -> ReLU -> Linear(256, class_num) on the decoder's hidden states.
    class_num >= 5 (the velocity task) swaps the decoder's input embedding for a class-label embedding of width 64 and its own
    Linear(64, d) (PianoBart.change_decoder_embedding, model.py:242-245); input_ids_decoder is then (B, S) labels."""

    def __init__(self, pianobart, class_num, hs, d_model=64):
        super().__init__()
        self.pianobart = pianobart
        if class_num >= 5:
            new_embedding = Embeddings(n_token=class_num, d_model=d_model)
            new_linear = nn.Linear(d_model, pianobart.bartConfig.d_model)
            self.pianobart.change_decoder_embedding(new_embedding, new_linear)
        self.classifier = nn.Sequential(nn.Dropout(0.1), nn.Linear(hs, 256), nn.ReLU(), nn.Linear(256, class_num))

    def forward(self, input_ids_encoder, input_ids_decoder, encoder_attention_mask=None, decoder_attention_mask=None):
        from . import heads
        x = self.pianobart(input_ids_encoder, input_ids_decoder, encoder_attention_mask, decoder_attention_mask).last_hidden_state
        c = self.classifier
        y = heads.dropout(x, c[0].p, self.training)
        y = heads.act(heads.linear(y, c[1].weight, c[1].bias), 'relu')
        return heads.linear(y, c[3].weight, c[3].bias)

