"""CPU: Octuple <-> MIDI codec against G13 -- vectors produced by the reference's own MIDI_to_encoding /
encoding_to_MIDI / padding (oracle/make_goldens.py g13) -- plus file round trips through the built-in SMF reader / writer and the
demo.py surface (Midi2Octuple / Octuple2Midi)."""
import os

import numpy as np
import pytest
import torch

from pianobart_amd import octuple_midi as om
from tests.golden_util import GOLD

Z = np.load(os.path.join(GOLD, 'g13_octuple_midi.npz'))


def _song():
    return om.Song(int(Z['ticks_per_beat']), [tuple(int(v) for v in n[:5]) + (bool(n[5]),) for n in Z['notes']],
                   [tuple(int(v) for v in c) for c in Z['ts']], [(int(c[0]), float(c[1])) for c in Z['tp']])


def test_tables_match_reference():
    t = Z['tables']
    assert [len(om.TS_LIST), len(om.DUR_ENC), len(om.DUR_DEC), om.ts_to_code(6, 8), om.dur_to_code(1000), om.code_to_dur(77), om.tempo_to_code(133.7)] == t.tolist()
    assert om.PAD_ROW == (256, 128, 129, 256, 128, 32, 254, 49) and om.EOS_ROW == (259, 131, 132, 259, 131, 35, 257, 52)
    assert om.ts_to_code(12, 4) == om.ts_to_code(6, 4) and om.ts_to_code(14, 128) == om.ts_to_code(7, 64)        # decomposition / reduction
    for bad in ((5, 3), (7, 128)):                                                 # outside the vocabulary: the reference asserts
        with pytest.raises(ValueError):
            om.ts_to_code(*bad)


def test_midi_to_encoding_matches_reference():
    enc = om.midi_to_encoding(_song())
    assert np.array_equal(np.array(enc, dtype=np.int64), Z['encoding'])
    assert om.midi_to_encoding(om.Song(480)) == []


def test_padding_matches_reference():
    enc = [tuple(r) for r in Z['encoding'].tolist()]
    assert np.array_equal(np.array(om.padding(enc[:300], 1024)), Z['padded_1024'])
    assert np.array_equal(np.array(om.padding(enc, 256, last=False)), Z['padded_cut_head'])
    assert np.array_equal(np.array(om.padding(enc, 256, last=True)), Z['padded_cut_tail'])
    assert om.padding(enc[:256], 256) == enc[:256]                                  # exactly full: neither PAD nor EOS


def test_encoding_to_midi_matches_reference():
    song = om.encoding_to_midi(Z['melodic'])
    notes = np.array(sorted([n[0], n[1], n[2], n[3], n[4], int(n[5])] for n in song.notes), dtype=np.int64)
    assert np.array_equal(notes, Z['back_notes'])
    assert np.array_equal(np.array(song.time_signatures, dtype=np.int64), Z['back_ts'])
    assert np.array_equal(np.array([c[0] for c in song.tempos]), Z['back_tp'][:, 0].astype(np.int64))
    assert np.allclose([c[1] for c in song.tempos], Z['back_tp'][:, 1], rtol=1e-12)


def test_file_round_trip_and_demo_surface(tmp_path):
    """rows -> Song -> .mid -> Song -> rows is the identity for placeable rows; Octuple2Midi cuts at the first special row like
    demo.py:72-99 and Midi2Octuple pads to the model's window."""
    rows, busy = [], {}
    for r in sorted(tuple(r) for r in Z['melodic'].tolist()):                        # a MIDI channel cannot hold two overlapping notes of one pitch
        start = r[0] * 10000 + r[1]
        if busy.get((r[2], r[3]), -1) <= start:
            rows.append(r)
            busy[(r[2], r[3])] = start + 10000 * 3
    assert len(rows) > 200
    song = om.encoding_to_midi(rows)
    path = str(tmp_path / 'a.mid')
    om.write_midi(song, path)
    back = om.read_midi(path)
    assert back.ticks_per_beat == 480 and sorted(back.notes) == sorted(song.notes)
    assert back.time_signatures == song.time_signatures and [c[0] for c in back.tempos] == [c[0] for c in song.tempos]
    again = om.midi_to_encoding(back)
    # a row's time-signature / tempo codes come back as its bar's majority / its position's mean (what the decoder wrote)
    assert [r[:6] for r in again] == [r[:6] for r in sorted(rows)]
    assert om.midi_to_encoding(om.encoding_to_midi(again)) == again                 # fixed point after one pass
    gen = torch.tensor(om.padding(rows[:100], 1024)).reshape(1, 1024, 8)
    gen[0, 40] = torch.tensor(om.EOS_ROW)                                           # generation stopped here
    out = str(tmp_path / 'gen.mid')
    assert om.Octuple2Midi(gen, out) and len(om.read_midi(out).notes) == 40
    oct2 = om.Midi2Octuple(out)
    assert tuple(oct2.shape) == (1, 1024, 8) and oct2.dtype == torch.int32
    assert [tuple(r) for r in oct2[0, :40, :6].tolist()] == [r[:6] for r in sorted(rows[:40])]
    assert (oct2[0, 40:] == torch.tensor(om.PAD_ROW)).all()
    assert not om.Octuple2Midi(torch.tensor([om.PAD_ROW] * 1024).reshape(1, 1024, 8), out)      # nothing generated
    drums = gen.clone(); drums[0, 5, 3] = 200                                        # percussion-range pitch ends the piece too
    assert om.octuple_to_rows(drums) == [tuple(r) for r in gen[0, :5].tolist()]


def test_reader_on_the_spec_built_byte_fixture_gives_the_reference_rows():
    """tests/golden/g13_song.mid was written byte by byte from the SMF specification (tests/smf_spec.py, make_smf_fixture.py: running
    status, note-on velocity 0 as note-off, sysex, controllers, pitch bend, text metas, five note tracks) and holds the G13 song:
    the reader must return that song and `midi_to_encoding` the 400 rows the REFERENCE's converter produced for it (demo.py:61-68)."""
    from tests import smf_spec
    path = os.path.join(GOLD, 'g13_song.mid')
    fmt, div, tracks = smf_spec.walk(open(path, 'rb').read())
    assert (fmt, div, len(tracks)) == (1, 384, 6)
    song = om.read_midi(path)
    assert song.ticks_per_beat == 384
    assert sorted(song.notes) == sorted(tuple(int(v) for v in n[:5]) + (bool(n[5]),) for n in Z['notes'])
    assert song.time_signatures == [tuple(int(v) for v in c) for c in Z['ts']]
    assert [c[0] for c in song.tempos] == Z['tp'][:, 0].astype(np.int64).tolist()
    assert np.allclose([c[1] for c in song.tempos], Z['tp'][:, 1], rtol=2e-6)           # a tempo is stored as whole microseconds per beat
    assert np.array_equal(np.array(om.midi_to_encoding(song), dtype=np.int64), Z['encoding'])
    window = om.Midi2Octuple(path, window=256)                                         # demo.py:61-68: tail of the piece + EOS
    assert np.array_equal(window[0].numpy().astype(np.int64), Z['padded_cut_tail'])


def test_writer_output_is_a_well_formed_smf_and_survives_the_spec_walker(tmp_path):
    """`write_midi` -> the independent strict walker (chunk sizes, data bytes, End of Track) -> the same song; and the fixture read,
    written and read again is unchanged where a file can say it (no two sounding notes of one pitch in a channel)."""
    from tests import smf_spec
    song = om.read_midi(os.path.join(GOLD, 'g13_song.mid'))
    keep, busy = [], {}
    for n in sorted(song.notes):
        if busy.get((n[4], n[5], n[2]), -1) <= n[0]:
            keep.append(n)
            busy[(n[4], n[5], n[2])] = n[1]
    assert len(keep) > 350
    song.notes = keep
    out = str(tmp_path / 'w.mid')
    om.write_midi(song, out)
    fmt, div, tracks = smf_spec.walk(open(out, 'rb').read())
    assert (fmt, div, len(tracks)) == (1, 384, 4)                                       # conductor + piano + program 40 + drums
    metas = [(t, d[0], bytes(d[1])) for t, s, d in tracks[0] if s == 0xFF]
    assert [(t, b[0], 2 ** b[1]) for t, k, b in metas if k == 0x58] == song.time_signatures
    assert [(t, int.from_bytes(b, 'big')) for t, k, b in metas if k == 0x51] == [(t, int(round(60000000 / bpm))) for t, bpm in song.tempos]
    ons = sorted((t, s & 15, d[0], d[1]) for tr in tracks[1:] for t, s, d in tr if s & 0xF0 == 0x90 and d[1] > 0)
    assert len(ons) == len(keep) and {c for _, c, _, _ in ons} == {0, 1, 9}
    back = om.read_midi(out)
    assert sorted(back.notes) == sorted(keep) and back.time_signatures == song.time_signatures
    assert np.array_equal(np.array(om.midi_to_encoding(back)), np.array(om.midi_to_encoding(song)))


def test_one_note_off_ends_every_earlier_note_of_its_pitch(tmp_path):
    """miditoolkit's pairing rule (the parser demo.py reads files with): a note-off closes all sounding notes of that channel and pitch
    begun at earlier ticks and leaves one begun at the same tick open."""
    from tests import smf_spec as S
    ev = [(0, 0, bytes([0x90, 60, 100])), (10, 0, bytes([0x90, 60, 90])), (20, 0, bytes([0x80, 60, 0])),
          (30, 1, bytes([0x90, 60, 80])), (30, 0, bytes([0x80, 60, 0])), (40, 0, bytes([0x80, 60, 0])), (50, 0, bytes([0x80, 60, 0]))]
    p = str(tmp_path / 'o.mid')
    open(p, 'wb').write(S.smf(96, [S.track(ev)], fmt=0))
    assert om.read_midi(p).notes == [(0, 20, 60, 100, 0, False), (10, 20, 60, 90, 0, False), (30, 40, 60, 80, 0, False)]
