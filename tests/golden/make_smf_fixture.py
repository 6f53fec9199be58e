"""Writes tests/golden/g13_song.mid: the G13 song (the notes, time-signature and tempo changes the reference's MIDI_to_encoding
was run on, g13_octuple_midi.npz) as a Standard MIDI File, byte by byte from the SMF specification through tests/smf_spec.py --
NOT through pianobart_amd.octuple_midi.write_midi, so that the product's reader is checked against an independent encoding:

  * format 1, division 384: a conductor track (track name, key signature, the 4 time signatures, the 4 tempo changes) and one
    track per (program, drum) lane; a lane never holds two overlapping notes of one pitch (the one thing an SMF cannot say
    unambiguously), so notes that overlap go to a further track of the same program on a channel of its own;
  * running status throughout; lane 0 ends its notes with note-on velocity 0, the others with note-off (release velocity 64);
  * events a reader must step over: track-name / text / marker metas, a sysex, controllers, channel pressure, pitch bend,
    polyphonic key pressure, a program change in the middle of nothing.

Run from the repo root: python tests/golden/make_smf_fixture.py   (deterministic; the .mid is committed).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import smf_spec as S                                                   # noqa: E402


def lanes_of(notes):
    """Greedy split of one instrument's notes into lanes without same-pitch overlap."""
    lanes = []
    for n in sorted(notes):
        for lane in lanes:
            if all(m[2] != n[2] or m[1] <= n[0] for m in lane):
                lane.append(n)
                break
        else:
            lanes.append([n])
    return lanes


def main():
    Z = np.load(os.path.join(HERE, 'g13_octuple_midi.npz'))
    div = int(Z['ticks_per_beat'])
    cond = [(0, 0, S.meta(0x03, b'conductor')), (0, 1, S.meta(0x59, [0xFE, 0]))]           # key signature: 2 flats, major
    for t, n, d in Z['ts'].tolist():
        cond.append((int(t), 2, S.time_signature(int(n), int(d))))
    for t, bpm in Z['tp'].tolist():
        cond.append((int(t), 3, S.tempo(float(bpm))))
    cond.append((div * 8, 4, S.meta(0x06, b'marker')))
    tracks = [S.track(cond)]
    groups = {}
    for s, e, p, v, prog, drum in Z['notes'].tolist():
        groups.setdefault((prog, drum), []).append((s, e, p, v))
    free = [c for c in range(16) if c != 9]
    k = 0
    for (prog, drum), notes in sorted(groups.items()):
        for lane in lanes_of(notes):
            ch = 9 if drum else free.pop(0)
            ev = [(0, 0, S.meta(0x03, ('lane %d' % k).encode())), (0, 1, bytes([0xC0 | ch, prog])),
                  (0, 2, bytes([0xB0 | ch, 7, 100])), (0, 2, bytes([0xB0 | ch, 10, 64]))]
            if k == 0:
                ev += [(1, 0, b'\xf0' + S.vlq(5) + bytes([0x7E, 0x7F, 0x09, 0x01, 0xF7])),   # GM system on
                       (div * 2, 0, bytes([0xB0 | ch, 64, 127])), (div * 3, 0, bytes([0xB0 | ch, 64, 0])),
                       (div * 4, 0, bytes([0xE0 | ch, 0, 0x50])), (div * 4 + 5, 0, bytes([0xD0 | ch, 33])),
                       (div * 5, 0, bytes([0xA0 | ch, 60, 20])), (div * 6, 0, S.meta(0x01, b'text'))]
            for s, e, p, v in lane:
                ev.append((s, 5, bytes([0x90 | ch, p, v])))
                ev.append((e, 4, bytes([0x90 | ch, p, 0]) if k == 0 else bytes([0x80 | ch, p, 64])))
            tracks.append(S.track(ev))
            k += 1
    raw = S.smf(div, tracks)
    S.walk(raw)
    with open(os.path.join(HERE, 'g13_song.mid'), 'wb') as f:
        f.write(raw)
    print('g13_song.mid: %d bytes, %d tracks (%d note lanes), %d notes' % (len(raw), len(tracks), k, len(Z['notes'])))


if __name__ == '__main__':
    main()
