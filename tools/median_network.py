"""what would EXACT 5x5 / 7x7 medians cost as separable sorting networks with shared columns (the formulation
DESIGN section 4's floor argument for k_median57 did not look at)?  Counted, not timed.

The generator builds min / max dataflow graphs for a strip of output pixels with every intermediate result hash-consed, so a sorted
column, a merged pair of sorted columns, a merged core of a tile of outputs ... is computed ONCE however many outputs use it (that is
the whole point of the formulation: Adams, "Fast median filters using separable sorting networks", SIGGRAPH 2021).  Pieces:

  sort_n       optimal sorting networks for n <= 8 (5, 9, 12, 16, 19 compare-exchanges for n = 4 .. 8)
  merge        Batcher's odd-even merge of two sorted lists of any lengths; `want` = the output positions needed: inputs that cannot
               reach them are dropped first (A[i] lands at merged positions i .. i+|B|), the rest is removed by dead-code elimination
  windows      sliding windows of K items, stride 1, shared hierarchically: aligned pairs -> cores of a tile of 2 outputs -> + 1 item
               (or one item per output: no sharing, for comparison)
  rank pruning an element at position i of a sorted list of s elements of the window (N = K*K, median rank m = (N-1)/2) is not the
               median if i > m or s-1-i > m: dropped, the rank still to be found shifts by the number dropped below

Every candidate scheme is CHECKED: random 8-bit images (and two-valued / constant / ramp ones) against numpy's median of the window.
The count is min/max operations per output pixel after dead-code elimination, taken as the marginal cost between two strip sizes (so
strip borders do not count), vertical phase (column sorts, shared between output rows) and horizontal phase separately.

usage: python tools/median_network.py            -> the table of profiles/r06_b_median.md
"""
import numpy as np

SORTERS = {      # optimal-size sorting networks (Knuth TAOCP 3, 5.3.4)
    1: [],
    2: [(0, 1)],
    3: [(0, 1), (1, 2), (0, 1)],
    4: [(0, 1), (2, 3), (0, 2), (1, 3), (1, 2)],
    5: [(0, 1), (3, 4), (2, 4), (2, 3), (1, 4), (0, 3), (0, 2), (1, 3), (1, 2)],
    6: [(1, 2), (4, 5), (0, 2), (3, 5), (0, 1), (3, 4), (2, 5), (0, 3), (1, 4), (2, 4), (1, 3), (2, 3)],
    7: [(1, 2), (3, 4), (5, 6), (0, 2), (3, 5), (4, 6), (0, 1), (4, 5), (2, 6), (0, 4), (1, 5), (0, 3), (2, 5), (1, 3), (2, 4), (2, 3)],
    8: [(0, 1), (2, 3), (4, 5), (6, 7), (0, 2), (1, 3), (4, 6), (5, 7), (1, 2), (5, 6), (0, 4), (3, 7), (1, 5), (2, 6), (1, 4), (3, 6),
        (2, 4), (3, 5), (3, 4)],
}


class Graph:
    """min / max nodes over input values, hash-consed.  Node ids: inputs are ('in', key); ops are ints."""

    def __init__(self):
        self.ops = []            # (kind, a, b)
        self.memo = {}
        self.inputs = {}
        self.cache = {}          # memo of composite constructions (sorted lists ...)

    def inp(self, key):
        if key not in self.inputs:
            self.inputs[key] = len(self.ops)
            self.ops.append(("in", key, None))
        return self.inputs[key]

    def op(self, kind, a, b):
        if a == b:
            return a
        if a > b:
            a, b = b, a
        k = (kind, a, b)
        if k not in self.memo:
            self.memo[k] = len(self.ops)
            self.ops.append(k)
        return self.memo[k]

    def ce(self, a, b):
        return self.op("min", a, b), self.op("max", a, b)

    def live(self, outs):
        seen = set()
        stack = list(outs)
        while stack:
            n = stack.pop()
            if n in seen:
                continue
            seen.add(n)
            kind, a, b = self.ops[n]
            if kind != "in":
                stack += [a, b]
        return seen

    def count(self, outs):
        return sum(1 for n in self.live(outs) if self.ops[n][0] != "in")

    def evaluate(self, outs, values):
        """values: dict input key -> numpy array (all the same shape).  Returns the arrays of `outs`."""
        live = sorted(self.live(outs))
        val = {}
        for n in live:
            kind, a, b = self.ops[n]
            val[n] = values[a] if kind == "in" else (np.minimum if kind == "min" else np.maximum)(val[a], val[b])
        return [val[n] for n in outs]


def sort_list(g, ids):
    ids = tuple(ids)
    key = ("sort", tuple(sorted(ids)))
    if key not in g.cache:
        v = list(sorted(ids))
        for (i, j) in SORTERS[len(v)]:
            v[i], v[j] = g.ce(v[i], v[j])
        g.cache[key] = tuple(v)
    return g.cache[key]


def _oddeven_merge(g, a, b):
    """Batcher's odd-even merge for two sorted lists of arbitrary lengths (recursive form on the odd / even subsequences)."""
    if not a:
        return list(b)
    if not b:
        return list(a)
    if len(a) == 1 and len(b) == 1:
        return list(g.ce(a[0], b[0]))
    if len(a) == 1 or len(b) == 1:           # insertion of one element: out_i = min(max(x, s[i-1]), s[i])
        x, s = (a[0], b) if len(a) == 1 else (b[0], a)
        out = [g.op("min", x, s[0])]
        for i in range(1, len(s)):
            out.append(g.op("min", g.op("max", x, s[i - 1]), s[i]))
        out.append(g.op("max", x, s[-1]))
        return out
    ev = _oddeven_merge(g, a[0::2], b[0::2])
    od = _oddeven_merge(g, a[1::2], b[1::2])
    # interleave: out[0] = ev[0]; then compare-exchange od[i] with ev[i+1]
    out = [ev[0]]
    i = 0
    while i < len(od) and i + 1 < len(ev):
        lo, hi = g.ce(od[i], ev[i + 1])
        out += [lo, hi]
        i += 1
    out += od[i:] + ev[i + 1:]
    return out


def merge(g, a, b, want=None):
    """Sorted merge of the sorted id lists a, b.  want = (lo, hi): only merged positions lo .. hi are needed -> returns exactly those
    (hi - lo + 1 ids); inputs that cannot reach them are dropped before the network is built."""
    a, b = list(a), list(b)
    n = len(a) + len(b)
    lo, hi = (0, n - 1) if want is None else want
    changed = True
    while changed:
        changed = False
        # elements that are certainly above position hi: A[i] with i > hi (at least i elements below it)
        while a and len(a) - 1 > hi:
            a.pop(); changed = True
        while b and len(b) - 1 > hi:
            b.pop(); changed = True
        # elements that are certainly below position lo: A[i] with i + |B| < lo  -> drop, positions shift down by one
        while a and len(b) < lo:          # i = 0
            a.pop(0); lo -= 1; hi -= 1; changed = True
        while b and len(a) < lo:
            b.pop(0); lo -= 1; hi -= 1; changed = True
    key = ("merge", tuple(a), tuple(b)) if tuple(a) <= tuple(b) else ("merge", tuple(b), tuple(a))
    if key not in g.cache:
        g.cache[key] = tuple(_oddeven_merge(g, a, b))
    full = g.cache[key]
    return list(full[lo:hi + 1])


# ---------------------------------------------------------------------------------------------------------------------------------
# vertical phase: sorted columns of K rows for every output row, shared between the output rows of a block

def column_sorted(g, x, y0, K, block):
    """sorted ids of rows y0 .. y0+K-1 of column x.  block = 1: a sort per output row; 2 / 4 / 8: output rows in aligned blocks.  The
    windows of a group of n consecutive output rows share the rows [first+n-1, first+K-1] (the group's core); halving the group adds
    n/2 rows above (first half) or below (second half) to the core: sorted on their own and merged in, down to the single window."""
    rows = lambda ys: [g.inp((y, x)) for y in ys]
    if block == 1:
        return sort_list(g, rows(range(y0, y0 + K)))

    def core(first, n):
        key = ("core", x, first, n, K)
        if key in g.cache:
            return g.cache[key]
        if n == block:
            r = sort_list(g, rows(range(first + n - 1, first + K)))
        else:
            parent_first = (first // (2 * n)) * (2 * n)
            pc = core(parent_first, 2 * n)
            if first == parent_first:            # first half: n rows above the parent's core
                extra = range(first + n - 1, first + 2 * n - 1)
            else:                                # second half: n rows below it
                extra = range(parent_first + K, parent_first + K + n)
            r = tuple(merge(g, pc, sort_list(g, rows(extra))))
        g.cache[key] = r
        return r

    return core(y0, 1)


def column7_from5(g, x, y0, block):
    """sorted 7-column of rows y0 .. y0+6 from the sorted 5-column of rows y0+1 .. y0+5 (which the 5x5 median needs anyway) + the two
    outer rows as a sorted pair."""
    s5 = column_sorted(g, x, y0 + 1, 5, block)
    pair = sort_list(g, [g.inp((y0, x)), g.inp((y0 + 6, x))])
    return tuple(merge(g, s5, pair))


# ---------------------------------------------------------------------------------------------------------------------------------
# horizontal phase: the median of K sorted columns, shared between neighbouring outputs

def median_independent(g, cols, N):
    """no sharing between outputs: a merge tree over the K sorted columns with rank pruning."""
    lists = [(list(c), len(c)) for c in cols]       # (kept ids, number of window elements the list stands for)
    m = (N - 1) // 2
    dropped_low = 0
    cur, cur_n = lists[0]
    for nxt, nxt_n in lists[1:]:
        s = len(cur) + len(nxt)
        # remaining problem: rank m - dropped_low among N_rem elements
        lo, hi = keep_in_remaining(s, N, m, dropped_low, cur_n + nxt_n)
        cur = merge(g, cur, nxt, (lo, hi))
        dropped_low += lo
        cur_n += nxt_n
    assert len(cur) == 1
    return cur[0]


def keep_in_remaining(s, N, m, dropped_low, covered):
    """merged list of s remaining elements; `covered` window elements are accounted for by it and the drops so far.  Positions that can
    hold the element of rank m - dropped_low of the remaining N - dropped elements."""
    # elements still outside this list: N - covered.  An element at position j has >= j smaller among the remaining and
    # >= s-1-j larger.  It can be the wanted one only if j <= r and (s-1-j) <= (N_rem - 1 - r), r = m - dropped_low.
    # N_rem = N - dropped_low - dropped_high; dropped_high is implied: covered = s + dropped_low + dropped_high
    dropped_high = covered - s - dropped_low
    n_rem = N - dropped_low - dropped_high
    r = m - dropped_low
    lo = max(0, s - 1 - (n_rem - 1 - r))
    hi = min(s - 1, r)
    return lo, hi


def median_tile2(g, colfn, x, K):
    """outputs in tiles of 2 (x even: the tile is x, x+1): windows [x-h, x+h] and [x-h+1, x+h+1] share K-1 columns = (K-1)/2 aligned
    pairs.  pair merges and the core are computed once per tile; each output adds one column."""
    h = K // 2
    N = K * K
    m = (N - 1) // 2
    t = x - (x % 2)                 # tile's first output
    core_cols = list(range(t - h + 1, t + h + 1))            # K-1 columns, starts at t-h+1
    # aligned pairs: (c, c+1) with c = core_cols[0], +2, ...   (alignment differs with h's parity, irrelevant for the count)
    pairs = [tuple(merge(g, colfn(c), colfn(c + 1))) for c in core_cols[0::2]]
    cur, covered, dropped_low = list(pairs[0]), 2 * K, 0
    for p in pairs[1:]:
        s = len(cur) + len(p)
        covered += 2 * K
        lo, hi = keep_in_remaining(s, N, m, dropped_low, covered)
        cur = merge(g, cur, p, (lo, hi))
        dropped_low += lo
    extra = colfn(t - h) if x == t else colfn(t + h + 1)
    s = len(cur) + K
    lo, hi = keep_in_remaining(s, N, m, dropped_low, covered + K)
    out = merge(g, cur, extra, (lo, hi))
    assert len(out) == 1
    return out[0]


def median_tile2_quads(g, colfn, x, K):
    """K = 7 only: as median_tile2, but the first two pairs of the core are merged into a quad that the NEXT-BUT-ONE tile's core shares
    ... (quads at every tile: core_t = merge(quad_t, pair_{t+2}); quad_t = merge(pair_t, pair_{t+1}) is also the tail of core_{t-1})."""
    assert K == 7
    h, N = 3, 49
    m = 24
    t = x - (x % 2)
    c0 = t - h + 1
    pair = lambda c: tuple(merge(g, colfn(c), colfn(c + 1)))
    # quad over columns c0 .. c0+3, alternatingly used as head (this tile) or tail (previous tile): choose by tile parity so that every
    # quad is used by two tiles
    if (t // 2) % 2 == 0:
        lo, hi = keep_in_remaining(28, N, m, 0, 28)
        quad = merge(g, pair(c0), pair(c0 + 2), (lo, hi))
        rest = pair(c0 + 4)
    else:
        lo, hi = keep_in_remaining(28, N, m, 0, 28)
        quad = merge(g, pair(c0 + 2), pair(c0 + 4), (lo, hi))
        rest = pair(c0)
    dropped_low = lo
    s = len(quad) + 14
    lo, hi = keep_in_remaining(s, N, m, dropped_low, 42)
    cur = merge(g, quad, rest, (lo, hi))
    dropped_low += lo
    extra = colfn(t - h) if x == t else colfn(t + h + 1)
    lo, hi = keep_in_remaining(len(cur) + 7, N, m, dropped_low, 49)
    out = merge(g, cur, extra, (lo, hi))
    assert len(out) == 1
    return out[0]


# ---------------------------------------------------------------------------------------------------------------------------------

def build(K, W, H, vblock, hscheme, col7from5=False, g=None):
    """a strip of W x H outputs (output (0,0) has its window's top-left input at (0,0))."""
    g = g or Graph()
    if K == 7 and col7from5:
        colf = lambda y: (lambda x: column7_from5(g, x, y, vblock))
    else:
        colf = lambda y: (lambda x: column_sorted(g, x, y, K, vblock))
    outs = {}
    h = K // 2
    for y in range(H):
        cf = colf(y)
        for x in range(W):
            # output x's window: input columns x .. x+K-1; schemes index columns by centre: centre = x + h
            xc = x + h
            if hscheme == "independent":
                outs[(y, x)] = median_independent(g, [cf(c) for c in range(xc - h, xc + h + 1)], K * K)
            elif hscheme == "tile2":
                outs[(y, x)] = median_tile2(g, cf, xc, K)
            elif hscheme == "tile2q":
                outs[(y, x)] = median_tile2_quads(g, cf, xc, K)
            else:
                raise ValueError(hscheme)
    return g, outs


def check(K, vblock, hscheme, col7from5=False, W=6, H=4, seed=0):
    g, outs = build(K, W, H, vblock, hscheme, col7from5)
    rng = np.random.default_rng(seed)
    T = 400
    keys = list(g.inputs)
    ys = [k[0] for k in keys]; xs = [k[1] for k in keys]
    y0, x0 = min(ys), min(xs)
    shape = (max(ys) - y0 + 1, max(xs) - x0 + 1)
    imgs = rng.integers(0, 256, (T,) + shape, dtype=np.int64)
    imgs[:50] = rng.integers(0, 2, (50,) + shape) * 255           # two-valued
    imgs[50:80] = rng.integers(0, 4, (30,) + shape)                # many ties
    imgs[80] = 7
    imgs[81] = np.arange(shape[0] * shape[1]).reshape(shape) % 256
    vals = {k: imgs[:, k[0] - y0, k[1] - x0] for k in keys}
    order = sorted(outs)
    res = g.evaluate([outs[o] for o in order], vals)
    h = K // 2
    for (y, x), r in zip(order, res):
        # window of output (y, x): rows y .. y+K-1, columns (x+h)-h .. (x+h)+h = x .. x+K-1 in input coordinates
        win = imgs[:, y - y0:y - y0 + K, x - x0:x - x0 + K].reshape(T, -1)
        ref = np.sort(win, axis=1)[:, (K * K - 1) // 2]
        if not (r == ref).all():
            return False
    return True


def marginal(K, vblock, hscheme, col7from5=False, with5=None):
    """min/max operations per output pixel: difference between strips of 16 x 8 and 8 x 8 ... both dimensions, per pixel.
    with5: also build the 5x5 medians in the SAME graph (shared column work when the 7-columns come from the 5-columns)."""
    def total(W, H):
        g, outs = build(K, W, H, vblock, hscheme, col7from5)
        o = list(outs.values())
        if with5 is not None:
            # 5x5 windows centred like the 7x7 ones: rows y+1 .. y+5, columns x+1 .. x+5
            for y in range(H):
                cf = lambda x, y=y: column_sorted(g, x, y + 1, 5, vblock)
                for x in range(W):
                    o.append(median_tile2(g, cf, x + 3, 5) if with5 == "tile2" else median_independent(g, [cf(c) for c in range(x + 1, x + 6)], 25))
        return g.count(o)
    # cost(W, H) ~ a*W*H + b*W + c*H + d  -> a from four strips
    a = (total(16, 8) - total(8, 8) - total(16, 4) + total(8, 4)) / (8 * 4)
    return a


def vertical_only(K, vblock, col7from5=False):
    def total(W, H):
        g = Graph()
        o = []
        for y in range(H):
            for x in range(W):
                o += list(column7_from5(g, x, y, vblock) if (K == 7 and col7from5) else column_sorted(g, x, y, K, vblock))
        return g.count(o)
    return (total(16, 8) - total(8, 8) - total(16, 4) + total(8, 4)) / (8 * 4)


def main():
    rows = []
    for K in (5, 7):
        for vblock in (1, 2, 4):
            v = vertical_only(K, vblock)
            for hs in ("independent", "tile2") + (("tile2q",) if K == 7 else ()):
                ok = check(K, vblock, hs)
                a = marginal(K, vblock, hs)
                rows.append((K, vblock, hs, False, v, a - v, a, ok))
    for vblock in (1, 2, 4):
        v = vertical_only(7, vblock, True)
        for hs in ("tile2", "tile2q"):
            ok = check(7, vblock, hs, True)
            a = marginal(7, vblock, hs, True)
            rows.append((7, vblock, hs, True, v, a - v, a, ok))
    print("| K | output rows per column block | horizontal scheme | 7-columns from 5-columns | column ops / pixel | merge + select ops / pixel | total min/max per pixel | exact vs numpy |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %d | %d | %s | %s | %.2f | %.2f | **%.2f** | %s |" % (r[0], r[1], r[2], "yes" if r[3] else "no", r[4], r[5], r[6], "yes" if r[7] else "NO"))
    print()
    for vblock in (2, 4):
        for hs7 in ("tile2", "tile2q"):
            both = marginal(7, vblock, hs7, True, with5="tile2")
            print("5x5 (tile2) + 7x7 (%s, 7-columns from the 5-columns), column blocks of %d, ONE graph: %.2f min/max per pixel" % (hs7, vblock, both))


if __name__ == "__main__":
    main()
