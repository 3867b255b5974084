"""Test-only: rewrites a baseline JPEG that Pillow encoded (one interleaved scan, standard Huffman tables) as the same
coefficients in SEVERAL NON-INTERLEAVED scans, one per component -- a legal sequential file that Pillow's encoder cannot
produce and libjpeg decodes to the very same pixels.  In such a scan the MCU is one block and only the blocks that hold image
samples are coded (T.81 A.2.2 / A.2.3), so the scan covers ceil(dw / 8) x ceil(dh / 8) blocks of the component in raster order
and its DC differences run along that order.  Pure Python, for images of a few hundred blocks."""
import struct

ZZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
      35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _segments(blob):
    """[(marker, payload offset, payload length)] up to and including SOS; returns them and the offset of the entropy data."""
    out, p = [], 2
    while True:
        assert blob[p] == 0xFF
        m = blob[p + 1]
        ln = struct.unpack(">H", blob[p + 2:p + 4])[0]
        out.append((m, p + 4, ln - 2))
        p += 2 + ln
        if m == 0xDA:
            return out, p


def _huff(counts, syms):
    dec, enc, code, k = {}, {}, 0, 0
    for ln in range(1, 17):
        for _ in range(counts[ln - 1]):
            dec[(ln, code)] = syms[k]
            enc[syms[k]] = (code, ln)
            code += 1
            k += 1
        code <<= 1
    return dec, enc


class _Bits:
    def __init__(self, data):
        self.d, self.p, self.acc, self.n = data, 0, 0, 0

    def bit(self):
        if self.n == 0:
            b = self.d[self.p]
            self.p += 1
            if b == 0xFF:
                assert self.d[self.p] == 0
                self.p += 1
            self.acc, self.n = b, 8
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k):
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v

    def sym(self, dec):
        code, ln = 0, 0
        while True:
            code = (code << 1) | self.bit()
            ln += 1
            if (ln, code) in dec:
                return dec[(ln, code)]
            assert ln < 16


def _extend(v, s):
    return v - (1 << s) + 1 if s and v < (1 << (s - 1)) else v


class _Out:
    def __init__(self):
        self.b, self.acc, self.n = bytearray(), 0, 0

    def put(self, code, ln):
        for i in range(ln - 1, -1, -1):
            self.acc = (self.acc << 1) | ((code >> i) & 1)
            self.n += 1
            if self.n == 8:
                self.b.append(self.acc)
                if self.acc == 0xFF:
                    self.b.append(0)
                self.acc, self.n = 0, 0

    def flush(self):
        while self.n:
            self.put(1, 1)                      # pad with 1-bits
        return bytes(self.b)


def _category(v):
    a, s = abs(v), 0
    while a:
        a >>= 1
        s += 1
    return s


def to_non_interleaved(blob, order=None):
    """blob: baseline JPEG, 3 components in one interleaved scan, no restart markers.  Returns the same image as three
    single-component scans, in `order` (component indices, default 0, 1, 2)."""
    segs, ent = _segments(blob)
    tabs, comps, X = {}, None, None
    for m, o, ln in segs:
        if m == 0xC4:
            i = o
            while i < o + ln:
                tc_th = blob[i]
                counts = list(blob[i + 1:i + 17])
                n = sum(counts)
                tabs[tc_th] = _huff(counts, list(blob[i + 17:i + 17 + n]))
                i += 17 + n
        elif m == 0xC0:
            Y, X, nc = struct.unpack(">HHB", blob[o + 1:o + 6])
            comps = [dict(id=blob[o + 6 + 3 * c], h=blob[o + 7 + 3 * c] >> 4, v=blob[o + 7 + 3 * c] & 15) for c in range(nc)]
        elif m == 0xDA:
            ns = blob[o]
            assert ns == 3 and comps is not None
            for k in range(ns):
                assert blob[o + 1 + 2 * k] == comps[k]["id"]
                comps[k]["td"], comps[k]["ta"] = blob[o + 2 + 2 * k] >> 4, blob[o + 2 + 2 * k] & 15
        else:
            assert m != 0xDD, "restart intervals are not handled here"
    hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
    mcux, mcuy = -(-X // (8 * hmax)), -(-Y // (8 * vmax))
    # decode: blocks[c][(row, col)] = 64 coefficients in zigzag order (DC absolute)
    rd = _Bits(blob[ent:])
    blocks = [dict() for _ in comps]
    pred = [0] * len(comps)
    for my in range(mcuy):
        for mx in range(mcux):
            for c, cc in enumerate(comps):
                dcd, acd = tabs[cc["td"]][0], tabs[0x10 | cc["ta"]][0]
                for by in range(cc["v"]):
                    for bx in range(cc["h"]):
                        z = [0] * 64
                        s = rd.sym(dcd)
                        pred[c] += _extend(rd.bits(s), s)
                        z[0] = pred[c]
                        k = 1
                        while k < 64:
                            rs = rd.sym(acd)
                            r, s = rs >> 4, rs & 15
                            if s:
                                k += r
                                z[k] = _extend(rd.bits(s), s)
                                k += 1
                            elif r == 15:
                                k += 16
                            else:
                                break
                        blocks[c][(my * cc["v"] + by, mx * cc["h"] + bx)] = z
    # everything up to the SOS marker is kept; three scans follow
    sos_at = segs[-1][1] - 4
    out = bytearray(blob[:sos_at])
    for c in (order or range(len(comps))):
        cc = comps[c]
        dw, dh = -(-X * cc["h"] // hmax), -(-Y * cc["v"] // vmax)
        dce, ace = tabs[cc["td"]][1], tabs[0x10 | cc["ta"]][1]
        out += b"\xff\xda" + struct.pack(">HB", 8, 1) + bytes([cc["id"], (cc["td"] << 4) | cc["ta"], 0, 63, 0])
        w = _Out()
        p = 0
        for row in range(-(-dh // 8)):
            for col in range(-(-dw // 8)):
                z = blocks[c][(row, col)]
                d = z[0] - p
                p = z[0]
                s = _category(d)
                w.put(*dce[s])
                if s:
                    w.put(d if d >= 0 else d + (1 << s) - 1, s)
                run = 0
                last = max([k for k in range(1, 64) if z[k]] or [0])
                for k in range(1, last + 1):
                    if z[k] == 0:
                        run += 1
                        continue
                    while run > 15:
                        w.put(*ace[0xF0])
                        run -= 16
                    s = _category(z[k])
                    w.put(*ace[(run << 4) | s])
                    w.put(z[k] if z[k] >= 0 else z[k] + (1 << s) - 1, s)
                    run = 0
                if last < 63:
                    w.put(*ace[0x00])
        out += w.flush()
    return bytes(out + b"\xff\xd9")


def _decode_baseline(blob):
    """(segments, comps, X, Y, tables, blocks): blocks[c][(row, col)] = 64 zigzag coefficients (DC absolute) of a baseline file
    with ONE scan (1 or 3 components, interleaved), no restart markers."""
    segs, ent = _segments(blob)
    tabs, comps, X, Y = {}, None, None, None
    for m, o, ln in segs:
        if m == 0xC4:
            i = o
            while i < o + ln:
                counts = list(blob[i + 1:i + 17])
                n = sum(counts)
                tabs[blob[i]] = _huff(counts, list(blob[i + 17:i + 17 + n]))
                i += 17 + n
        elif m == 0xC0:
            Y, X, nc = struct.unpack(">HHB", blob[o + 1:o + 6])
            comps = [dict(id=blob[o + 6 + 3 * c], h=blob[o + 7 + 3 * c] >> 4, v=blob[o + 7 + 3 * c] & 15) for c in range(nc)]
        elif m == 0xDA:
            assert blob[o] == len(comps)
            for k in range(blob[o]):
                comps[k]["td"], comps[k]["ta"] = blob[o + 2 + 2 * k] >> 4, blob[o + 2 + 2 * k] & 15
        else:
            assert m != 0xDD
    hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
    mcux, mcuy = -(-X // (8 * hmax)), -(-Y // (8 * vmax))
    rd = _Bits(blob[ent:])
    blocks = [dict() for _ in comps]
    pred = [0] * len(comps)
    for my in range(mcuy):
        for mx in range(mcux):
            for c, cc in enumerate(comps):
                dcd, acd = tabs[cc["td"]][0], tabs[0x10 | cc["ta"]][0]
                for by in range(cc["v"]):
                    for bx in range(cc["h"]):
                        z = [0] * 64
                        s = rd.sym(dcd)
                        pred[c] += _extend(rd.bits(s), s)
                        z[0] = pred[c]
                        k = 1
                        while k < 64:
                            rs = rd.sym(acd)
                            r, s = rs >> 4, rs & 15
                            if s:
                                k += r
                                z[k] = _extend(rd.bits(s), s)
                                k += 1
                            elif r == 15:
                                k += 16
                            else:
                                break
                        blocks[c][(my * cc["v"] + by, mx * cc["h"] + bx)] = z
    return segs, comps, X, Y, tabs, blocks


def to_progressive(blob, script, overrun=None):
    """blob: a baseline JPEG Pillow wrote (one interleaved scan, standard Huffman tables).  Returns a PROGRESSIVE file (SOF2) holding
    the same coefficients in first passes only (Ah = Al = 0): the interleaved DC scan, then the AC scans of `script`, a list of
    (component, ss, se, scale): `scale` multiplies the coefficients this scan codes (1 = the image's; -1 codes their negatives, which
    makes a scan that REPEATS a band carry different values -- a non-conforming but accepted stream, libjpeg applies the scans in
    file order).  overrun = (scan index in `script`, value): in that scan every block whose band ends in zeros gets one more code
    word whose run carries PAST se (libjpeg stores it all the same, jdphuff.c) -- into a band a later scan also writes."""
    segs, comps, X, Y, tabs, blocks = _decode_baseline(blob)
    hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
    mcux, mcuy = -(-X // (8 * hmax)), -(-Y // (8 * vmax))
    sos_at = segs[-1][1] - 4
    out = bytearray(blob[:sos_at])
    for m, o, ln in segs:                                     # SOF0 -> SOF2
        if m == 0xC0:
            out[o - 3] = 0xC2
    # DC first pass, interleaved
    out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(comps), len(comps))
    for cc in comps:
        out += bytes([cc["id"], cc["td"] << 4])
    out += bytes([0, 0, 0])
    w = _Out()
    pred = [0] * len(comps)
    for my in range(mcuy):
        for mx in range(mcux):
            for c, cc in enumerate(comps):
                dce = tabs[cc["td"]][1]
                for by in range(cc["v"]):
                    for bx in range(cc["h"]):
                        z = blocks[c][(my * cc["v"] + by, mx * cc["h"] + bx)]
                        d = z[0] - pred[c]
                        pred[c] = z[0]
                        s = _category(d)
                        w.put(*dce[s])
                        if s:
                            w.put(d if d >= 0 else d + (1 << s) - 1, s)
    out += w.flush()
    for si, (c, ss, se, scale) in enumerate(script):
        cc = comps[c]
        dw, dh = -(-X * cc["h"] // hmax), -(-Y * cc["v"] // vmax)
        ace = tabs[0x10 | cc["ta"]][1]
        out += b"\xff\xda" + struct.pack(">HB", 8, 1) + bytes([cc["id"], cc["ta"], ss, se, 0])
        w = _Out()
        for row in range(-(-dh // 8)):
            for col in range(-(-dw // 8)):
                z = blocks[c][(row, col)]
                run, k_next = 0, ss
                last = max([k for k in range(ss, se + 1) if z[k]] or [ss - 1])
                for k in range(ss, last + 1):
                    v = z[k] * scale
                    if v == 0:
                        run += 1
                        continue
                    while run > 15:
                        w.put(*ace[0xF0])
                        run -= 16
                    s = _category(v)
                    w.put(*ace[(run << 4) | s])
                    w.put(v if v >= 0 else v + (1 << s) - 1, s)
                    run = 0
                    k_next = k + 1
                if overrun is not None and overrun[0] == si and last < se and se + 2 - max(last + 1, ss) <= 15 and se + 2 <= 63:
                    v = overrun[1]                                     # lands on coefficient se + 2 and ends the block (k > Se)
                    s = _category(v)
                    w.put(*ace[((se + 2 - max(last + 1, ss)) << 4) | s])
                    w.put(v if v >= 0 else v + (1 << s) - 1, s)
                elif last < se:
                    w.put(*ace[0x00])                                  # EOB0
        out += w.flush()
    return bytes(out + b"\xff\xd9")
