"""Octuple <-> MIDI, dependency-free: the pre/post-processing either side of generate.

Reference: `Midi2Octuple` / `Octuple2Midi` (demo.py:61-102) on top of `MIDI_to_encoding`, `encoding_to_MIDI` and `padding`
(Data/data_generation/convert.py:157-333), which need the third-party `miditoolkit` (absent from this image). Here the same
conversion works on a small in-memory `Song` (notes, time-signature and tempo changes in ticks) and a minimal Standard MIDI
File reader / writer, so a generated (1, 1024, 8) tensor can be turned into a playable .mid and a .mid into the model's input
with nothing but the standard library.

Octuple row = (bar, position, program, pitch, duration, velocity, time signature, tempo), classes order of PianoBart.py:29:
  * position: 1/16 of a beat (pos_resolution 16); bar / position follow the running time signature;
  * program 128 = percussion, whose pitches are stored + 128;
  * duration: index into a piecewise-linear / exponential table (16 steps per octave of length); velocity: // 4;
  * time signature: index into the (numerator, denominator) list, denominators 1..64, at most 2 whole notes per bar;
  * tempo: round(12 * log2(bpm / 16)), bpm clipped to [16, 256].
Special rows (convert.py:42-43,321-333): PAD = token_boundary + 1, EOS = token_boundary + 4.
"""
import math
import struct

import numpy as np

POS_RESOLUTION = 16          # positions per beat
MAX_TS_DENOM_POW = 6         # x/1 .. x/64
MAX_NOTES_PER_BAR = 2        # in whole notes
TRUNC_POS = 2 ** 16
TOKEN_BOUNDARY = (255, 127, 128, 255, 127, 31, 253, 48)
PAD_ROW = tuple(b + 1 for b in TOKEN_BOUNDARY)
EOS_ROW = tuple(b + 4 for b in TOKEN_BOUNDARY)

# time-signature vocabulary: every (n, 2^i) with n <= 2 * 2^i, in the order convert.py:83-88 enumerates it
TS_LIST = [(n, 2 ** i) for i in range(MAX_TS_DENOM_POW + 1) for n in range(1, 2 ** i * MAX_NOTES_PER_BAR + 1)]
TS_INDEX = {ts: k for k, ts in enumerate(TS_LIST)}


def _duration_tables(octaves=8):
    """Duration code -> length in positions and back (convert.py:89-95): octave i holds 16 codes, each 2^i positions wide."""
    dec, enc = [], []
    for i in range(octaves):
        for _ in range(POS_RESOLUTION):
            dec.append(len(enc))
            enc.extend([len(dec) - 1] * (2 ** i))
    return enc, dec


DUR_ENC, DUR_DEC = _duration_tables()


def ts_to_code(num, den):
    """Reduce a time signature into the vocabulary (convert.py:131-143) and return its index."""
    while den > 2 ** MAX_TS_DENOM_POW and den % 2 == 0 and num % 2 == 0:
        den //= 2
        num //= 2
    while num > MAX_NOTES_PER_BAR * den:
        num //= next(f for f in range(2, num + 1) if num % f == 0)
    if (num, den) not in TS_INDEX:
        raise ValueError('unsupported time signature: %s' % ((num, den),))
    return TS_INDEX[(num, den)]


def dur_to_code(npos):
    return DUR_ENC[npos] if npos < len(DUR_ENC) else DUR_ENC[-1]


def code_to_dur(code):
    return DUR_DEC[code] if code < len(DUR_DEC) else DUR_DEC[-1]


def tempo_to_code(bpm):
    return round(math.log2(min(max(bpm, 16), 256) / 16) * 12)


def code_to_tempo(code):
    return 2 ** (code / 12) * 16


def bar_length(ts_code):
    n, d = TS_LIST[ts_code]
    return n * 4 * POS_RESOLUTION // d


class Song:
    """notes: (start, end, pitch, velocity, program, is_drum) in ticks; time_signatures: (tick, numerator, denominator);
    tempos: (tick, bpm)."""

    def __init__(self, ticks_per_beat=480, notes=None, time_signatures=None, tempos=None):
        self.ticks_per_beat = ticks_per_beat
        self.notes = list(notes or [])
        self.time_signatures = list(time_signatures or [])
        self.tempos = list(tempos or [])


# ---------------------------------------------------------------------------------------------------- MIDI -> Octuple
def midi_to_encoding(song):
    """convert.py:157-244 (task 'pretrain'): a sorted list of 8-tuples, one per note."""
    to_pos = lambda t: round(t * POS_RESOLUTION / song.ticks_per_beat)
    starts = [to_pos(n[0]) for n in song.notes]
    if not starts:
        return []
    npos = min(max(starts) + 1, TRUNC_POS)
    ts_at = np.full(npos, ts_to_code(4, 4), dtype=np.int64)          # MIDI defaults: 4/4, 120 bpm
    tp_at = np.full(npos, tempo_to_code(120.0), dtype=np.int64)
    for changes, track, code in ((sorted(song.time_signatures), ts_at, lambda c: ts_to_code(c[1], c[2])),
                                 (sorted(song.tempos), tp_at, lambda c: tempo_to_code(c[1]))):
        for k, c in enumerate(changes):
            lo = to_pos(c[0])
            hi = to_pos(changes[k + 1][0]) if k + 1 < len(changes) else npos
            track[max(lo, 0):max(min(hi, npos), 0)] = code(c)
    bar_of, pos_of = np.zeros(npos, dtype=np.int64), np.zeros(npos, dtype=np.int64)
    bar = cnt = 0
    length = None
    for j in range(npos):                                            # the running time signature is latched at each bar line
        if cnt == 0:
            length = bar_length(int(ts_at[j]))
        bar_of[j], pos_of[j] = bar, cnt
        cnt += 1
        if cnt >= length:
            cnt -= length
            bar += 1
    rows = []
    for (start, end, pitch, vel, program, is_drum), p in zip(song.notes, starts):
        if p >= TRUNC_POS:
            continue
        rows.append((int(bar_of[p]), int(pos_of[p]), 129 if is_drum else program, pitch + 256 if is_drum else pitch,
                     dur_to_code(to_pos(end) - p), vel // 4, int(ts_at[p]), int(tp_at[p])))
    rows.sort()
    return rows


def padding(rows, window=1024, last=False):
    """convert.py:321-333: PAD rows up to `window`; an over-long piece keeps window-1 rows (its tail if `last`) + an EOS row."""
    rows = list(rows)
    if len(rows) > window:
        rows = rows[1 - window:] if last else rows[:window - 1]
        return rows + [EOS_ROW]
    return rows + [PAD_ROW] * (window - len(rows))


# ---------------------------------------------------------------------------------------------------- Octuple -> MIDI
def encoding_to_midi(rows, ticks_per_beat=480):
    """convert.py:248-319: a Song from note rows (no special rows). Bars take their most frequent time signature, positions the
    rounded mean of their tempo codes; gaps inherit from the left."""
    rows = [tuple(int(v) for v in r) for r in rows]
    nbars = max(r[0] for r in rows) + 1
    votes = [[] for _ in range(nbars)]
    for r in rows:
        votes[r[0]].append(r[6])
    bar_ts, prev = [], None
    for k, v in enumerate(votes):
        cur = max(set(v), key=v.count) if v else (ts_to_code(4, 4) if k == 0 else prev)
        bar_ts.append(cur)
        prev = cur
    bar_start, cur = [], 0
    for code in bar_ts:
        bar_start.append(cur)
        if 0 <= code < len(TS_LIST):
            cur += bar_length(code)
    npos = cur + max(r[1] for r in rows)
    tempo_votes = [[] for _ in range(npos)]
    for r in rows:
        p = bar_start[r[0]] + r[1]
        if p < npos:
            tempo_votes[p].append(r[7])
    pos_tp, prev = [], None
    for k, v in enumerate(tempo_votes):
        prev = round(sum(v) / len(v)) if v else (tempo_to_code(120.0) if k == 0 else prev)
        pos_tp.append(prev)
    tick = lambda bar, pos: (bar_start[bar] + pos) * ticks_per_beat // POS_RESOLUTION
    song = Song(ticks_per_beat)
    for bar, pos, program, pitch, dur, vel, _, _ in rows:
        if not 0 <= program <= 128:
            continue
        start = tick(bar, pos)
        length = max(1, tick(0, code_to_dur(dur)))
        song.notes.append((start, start + length, pitch - 128 if program == 128 else pitch, vel * 4 + 2, 0 if program == 128 else program,
                           program == 128))
    cur = None
    for k, code in enumerate(bar_ts):
        if code != cur and 0 <= code < len(TS_LIST):
            song.time_signatures.append((tick(k, 0),) + TS_LIST[code])
            cur = code
    cur = None
    for k, code in enumerate(pos_tp):
        if code != cur:
            song.tempos.append((tick(0, k), code_to_tempo(code)))
            cur = code
    return song


def octuple_to_rows(octuple):
    """demo.py:72-99: cut a generated (S, 8) / (1, S, 8) array at its first special or percussion-range row (which becomes the EOS),
    or at its last row; returns the note rows in front of it (None when there is none: 'Generate Fail! (empty)')."""
    if hasattr(octuple, 'detach'):
        octuple = octuple.detach().cpu().numpy()
    a = np.array(octuple).reshape(-1, 8).astype(np.int64)
    bad = (a >= np.array(PAD_ROW)).any(axis=1) | (a[:, 3] > 127)
    end = int(np.argmax(bad)) if bad.any() else len(a) - 1
    return [tuple(int(v) for v in r) for r in a[:end]] if end > 0 else None


# ---------------------------------------------------------------------------------------------------- Standard MIDI File I/O
def _vlq(n):
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def write_midi(song, path):
    """Format-1 SMF: a conductor track (time signatures, tempos) + one track per (program, is_drum)."""
    def track(events):
        data, last = bytearray(), 0
        for t, _, payload in sorted(events, key=lambda e: (e[0], e[1])):
            data += _vlq(t - last) + payload
            last = t
        data += b'\x00\xff\x2f\x00'
        return b'MTrk' + struct.pack('>I', len(data)) + bytes(data)

    cond = []
    for t, n, d in song.time_signatures:
        cond.append((t, 0, bytes([0xFF, 0x58, 4, n, int(math.log2(d)), 24, 8])))
    for t, bpm in song.tempos:
        cond.append((t, 1, b'\xff\x51\x03' + struct.pack('>I', int(round(60000000 / bpm)))[1:]))
    tracks = [track(cond)]
    groups = {}
    for n in song.notes:
        groups.setdefault((n[4], n[5]), []).append(n)
    melodic = 0
    for (program, is_drum), notes in sorted(groups.items()):
        if is_drum:
            ch = 9
        else:
            ch = melodic % 15
            ch += ch >= 9
            melodic += 1
        ev = [(0, 0, bytes([0xC0 | ch, program & 0x7F]))]
        for start, end, pitch, vel, _, _ in notes:
            ev.append((start, 2, bytes([0x90 | ch, pitch & 0x7F, max(1, min(127, vel))])))
            ev.append((end, 1, bytes([0x80 | ch, pitch & 0x7F, 0])))
        tracks.append(track(ev))
    with open(path, 'wb') as f:
        f.write(b'MThd' + struct.pack('>IHHH', 6, 1, len(tracks), song.ticks_per_beat) + b''.join(tracks))


def read_midi(path):
    """Notes (note-on/off per track, channel and pitch; running status; note-on with velocity 0 = note-off), program changes, tempo
    and time-signature metas; every other event is stepped over. Pinned by tests/golden/g13_song.mid, a byte fixture built from the
    SMF specification without this module (tests/golden/make_smf_fixture.py): reading it gives the G13 song and, through
    `midi_to_encoding`, the reference's own 400 rows."""
    raw = open(path, 'rb').read()
    if raw[:4] != b'MThd':
        raise ValueError('%s is not a Standard MIDI File' % path)
    _, _, ntrk, div = struct.unpack('>IHHH', raw[4:14])
    if div & 0x8000:
        raise ValueError('SMPTE time division is not supported')
    song, off = Song(div), 14
    for _ in range(ntrk):
        if raw[off:off + 4] != b'MTrk':
            raise ValueError('bad track header')
        size = struct.unpack('>I', raw[off + 4:off + 8])[0]
        p, stop = off + 8, off + 8 + size
        off = stop
        t, status, program, open_notes = 0, 0, {}, {}
        while p < stop:
            dt = 0
            while True:
                b = raw[p]; p += 1
                dt = (dt << 7) | (b & 0x7F)
                if not b & 0x80:
                    break
            t += dt
            b = raw[p]
            if b & 0x80:
                status = b; p += 1
            if status == 0xFF:
                kind = raw[p]; p += 1
                ln = 0
                while True:
                    b = raw[p]; p += 1
                    ln = (ln << 7) | (b & 0x7F)
                    if not b & 0x80:
                        break
                body = raw[p:p + ln]; p += ln
                if kind == 0x51 and ln == 3:
                    song.tempos.append((t, 60000000 / int.from_bytes(body, 'big')))
                elif kind == 0x58 and ln >= 2:
                    song.time_signatures.append((t, body[0], 2 ** body[1]))
            elif status in (0xF0, 0xF7):
                ln = 0
                while True:
                    b = raw[p]; p += 1
                    ln = (ln << 7) | (b & 0x7F)
                    if not b & 0x80:
                        break
                p += ln
            else:
                hi, ch = status & 0xF0, status & 0x0F
                if hi in (0xC0, 0xD0):
                    if hi == 0xC0:
                        program[ch] = raw[p]
                    p += 1
                else:
                    d1, d2 = raw[p], raw[p + 1]; p += 2
                    if hi == 0x90 and d2 > 0:
                        open_notes.setdefault((ch, d1), []).append((t, d2))
                    elif hi == 0x80 or (hi == 0x90 and d2 == 0):
                        # miditoolkit's rule (its parser is what demo.py:61-68 reads files with; restated from the published
                        # package, source absent here): ONE note-off ends EVERY sounding note of that track / channel / pitch
                        # that began at an earlier tick; a note begun at this very tick stays open.
                        pend = open_notes.get((ch, d1), [])
                        for start, vel in pend:
                            if start != t:
                                song.notes.append((start, t, d1, vel, 0 if ch == 9 else program.get(ch, 0), ch == 9))
                        open_notes[(ch, d1)] = [n for n in pend if n[0] == t]
    song.notes.sort()
    return song


# ---------------------------------------------------------------------------------------------------- demo.py surface
def Midi2Octuple(midi_path, window=1024):
    """demo.py:61-68: .mid -> (1, window, 8) int tensor (tail of the piece when it is longer than the window)."""
    import torch
    rows = padding(midi_to_encoding(read_midi(midi_path)), window=window, last=True)
    return torch.tensor([rows]).to(torch.int)


def Octuple2Midi(octuple, midi_path):
    """demo.py:72-102: generated tensor -> .mid (returns False and prints the reference's message when nothing was generated)."""
    import torch
    rows = octuple_to_rows(octuple.detach().cpu().numpy() if isinstance(octuple, torch.Tensor) else octuple)
    if not rows:
        print("Generate Fail! (empty)")
        return False
    write_midi(encoding_to_midi(rows), midi_path)
    return True
