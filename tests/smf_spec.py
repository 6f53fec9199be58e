"""A Standard MIDI File byte builder and a strict event walker written from the SMF 1.0 specification ("Standard MIDI-File
Format Spec. 1.1", MMA), sharing no code with `pianobart_amd/octuple_midi.py`: the builder makes the committed byte fixture
`tests/golden/g13_song.mid` (make_smf_fixture.py), the walker checks what `write_midi` emits. Test infrastructure only.

Chunk layout (spec section 2): `MThd` <u32 6> <u16 format> <u16 ntrks> <u16 division>, then ntrks x `MTrk` <u32 length> <events>.
Event (section 3): <variable-length delta time> then a MIDI channel message (status 0x8n..0xEn with 1 or 2 data bytes, status
omitted under RUNNING STATUS when it repeats the previous channel status), a sysex (F0 / F7 <vlq length> <bytes>) or a meta event
(FF <type> <vlq length> <bytes>); every track ends with FF 2F 00. Sysex and meta events cancel running status.
"""
import struct

DATA_BYTES = {0x80: 2, 0x90: 2, 0xA0: 2, 0xB0: 2, 0xC0: 1, 0xD0: 1, 0xE0: 2}


def vlq(n):
    """Variable-length quantity: 7 bits per byte, most significant first, bit 7 set on all but the last."""
    assert 0 <= n < 1 << 28
    groups = [n & 0x7F]
    while n > 0x7F:
        n >>= 7
        groups.append(0x80 | (n & 0x7F))
    return bytes(groups[::-1])


def meta(kind, body):
    return bytes([0xFF, kind]) + vlq(len(body)) + bytes(body)


def time_signature(num, den):
    """FF 58 04 nn dd cc bb: dd = log2 of the denominator, 24 MIDI clocks per metronome click, 8 32nd notes per quarter."""
    dd = den.bit_length() - 1
    assert 1 << dd == den
    return meta(0x58, [num, dd, 24, 8])


def tempo(bpm):
    """FF 51 03 tttttt: microseconds per quarter note."""
    return meta(0x51, int(round(60000000 / bpm)).to_bytes(3, 'big'))


def track(events, running_status=True):
    """events: (tick, order, payload) with payload = a complete event (status included). Channel messages whose status equals the
    previous channel message's are written without it when `running_status`."""
    out, last_tick, last_status = bytearray(), 0, None
    for tick, _, payload in sorted(events, key=lambda e: (e[0], e[1])):
        out += vlq(tick - last_tick)
        last_tick = tick
        st = payload[0]
        if st >= 0xF0:
            last_status = None
            out += payload
        elif running_status and st == last_status:
            out += payload[1:]
        else:
            last_status = st
            out += payload
    out += b'\x00' + meta(0x2F, b'')
    return b'MTrk' + struct.pack('>I', len(out)) + bytes(out)


def smf(division, tracks, fmt=1):
    return b'MThd' + struct.pack('>IHHH', 6, fmt, len(tracks), division) + b''.join(tracks)


def walk(raw):
    """Strict walk of an SMF: returns (format, division, [[(tick, status, data bytes or meta (kind, body))]]) and raises
    AssertionError on anything the specification does not allow (bad chunk sizes, data byte with bit 7 set, missing End of Track,
    bytes behind it, running status without a status)."""
    assert raw[:4] == b'MThd' and struct.unpack('>I', raw[4:8])[0] == 6
    fmt, ntrk, div = struct.unpack('>HHH', raw[8:14])
    assert fmt in (0, 1, 2) and (fmt != 0 or ntrk == 1)
    off, tracks = 14, []
    for _ in range(ntrk):
        assert raw[off:off + 4] == b'MTrk'
        size = struct.unpack('>I', raw[off + 4:off + 8])[0]
        p, stop = off + 8, off + 8 + size
        assert stop <= len(raw)
        off = stop
        tick, status, ev, ended = 0, None, [], False

        def read_vlq(p):
            n = 0
            for k in range(4):
                b = raw[p]; p += 1
                n = (n << 7) | (b & 0x7F)
                if not b & 0x80:
                    return n, p
            raise AssertionError('variable-length quantity longer than 4 bytes')

        while p < stop:
            assert not ended, 'event behind End of Track'
            dt, p = read_vlq(p)
            tick += dt
            b = raw[p]
            if b == 0xFF:
                kind = raw[p + 1]
                assert kind < 0x80
                ln, p = read_vlq(p + 2)
                ev.append((tick, 0xFF, (kind, raw[p:p + ln])))
                p += ln
                status = None
                ended = kind == 0x2F
                assert not ended or ln == 0
            elif b in (0xF0, 0xF7):
                ln, p = read_vlq(p + 1)
                ev.append((tick, b, raw[p:p + ln]))
                p += ln
                status = None
            else:
                if b & 0x80:
                    assert b < 0xF0, 'system common / real-time byte inside a file'
                    status = b
                    p += 1
                assert status is not None, 'data byte without running status'
                n = DATA_BYTES[status & 0xF0]
                data = raw[p:p + n]
                assert len(data) == n and all(d < 0x80 for d in data)
                ev.append((tick, status, bytes(data)))
                p += n
        assert ended and p == stop
        tracks.append(ev)
    assert off == len(raw), 'bytes behind the last chunk'
    return fmt, div, tracks
