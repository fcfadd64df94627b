"""Design probe (CPU, numpy; not a test): the late rounds of the rotation sort of a BZip2 block with and without the CHAIN shortcut of DESIGN.md 9.

Prefix doubling keeps, after the round for prefix h, the groups of rows that agree on their first h bytes; a round sorts every group by the class
of its rows' second halves (row + h).  On data with long repeats the groups sit on chains -- the group of the rows x1 .. xk is followed by the group
of x1 + 1 .. xk + 1 -- and the order inside a group is the order inside its successor: a whole chain can take the keys of its LAST group (whose
successors' classes differ).  This script runs both on one block of the benchmark stream, checks that they end in the same order, and prints the
rounds and the rows looked at per round.

    python tests/probes/bwt_chain_rounds.py [block bytes, default 900000] [offset into the stream, default 0]
"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _common import mixlib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 900000
off = int(sys.argv[2]) if len(sys.argv) > 2 else 0
M = mixlib()
t = np.zeros(n, np.uint8)
M.zada_silesia_mix(ctypes.c_uint64(0x5A1E51A), ctypes.c_uint32(0x1F), ctypes.c_uint64(off), ctypes.c_uint64(n), ctypes.c_void_p(t.ctypes.data))
t = t.astype(np.int64)
idx = np.arange(n)


def classes_of(order, key):
    k = key[order]
    head = np.ones(n, bool); head[1:] = k[1:] != k[:-1]
    first = np.maximum.accumulate(np.where(head, idx, 0))
    c = np.empty(n, np.int64); c[order] = first
    return c


# the swept rounds (whole array) up to prefix 8, as the product does before it switches to the lists
cls = classes_of(np.argsort(t, kind="stable"), t)
sa = np.argsort(cls, kind="stable")
h = 1
while h < 8:
    key = cls * (n + 1) + cls[(idx + h) % n]
    sa = np.argsort(key, kind="stable")
    cls = classes_of(sa, key)
    h *= 2


def run(chain):
    c = cls.copy(); order = sa.copy(); hh = h
    rounds, looked = 0, []
    while True:
        size = np.bincount(c, minlength=n)
        uns = size[c] > 1                                        # rows of groups of more than one row
        rows = np.nonzero(uns)[0]
        if len(rows) == 0 or hh >= 2 * n:
            break
        g = c[rows]                                              # the group of a row = its class (first index of the group in the sorted order)
        src = rows                                               # whose second half gives the row's key
        if chain:
            # rows of every group in position order; a group links to the group of its rows + 1 if that is the same rows + 1, member by member
            o = np.lexsort((rows, g)); rs, gs = rows[o], g[o]
            start = np.r_[True, gs[1:] != gs[:-1]]
            goff = np.maximum.accumulate(np.where(start, np.arange(len(rs)), 0))
            prank = np.arange(len(rs)) - goff                    # rank of the row inside its group, by position
            where = np.full(n, -1, np.int64); where[rs] = np.arange(len(rs))
            succ = (rs + 1) % n
            ws = where[succ]                                     # where the successor row sits in rs (-1: not in an unsorted group)
            ok = ws >= 0
            ok &= np.where(ok, prank[np.maximum(ws, 0)] == prank, False)
            sg = np.where(ok, gs[np.maximum(ws, 0)], -1)         # the successor's group
            first_sg = sg[goff]                                  # ... of the group's first row
            ok &= (sg == first_sg) & (sg != gs)
            ok &= size[np.maximum(sg, 0)] == size[gs]
            bad_groups = np.zeros(n, bool); bad_groups[gs[~ok]] = True
            link = np.full(n, -1, np.int64)                      # by group id (= class)
            heads = rs[start]; hg = gs[start]
            good = ~bad_groups[hg]
            link[hg[good]] = first_sg[start][good]
            term = np.where(link >= 0, link, np.arange(n))      # pointer jumping to the chain's last group
            dist = np.where(link >= 0, 1, 0)
            for _ in range(20):
                nt = term[term]
                nd = dist + dist[term]
                if np.array_equal(nt, term):
                    break
                term, dist = nt, nd
            # the row of the last group that corresponds to a row: same rank by position = the row + the distance along the chain
            src_sorted = (rs + dist[gs]) % n
            src = np.empty(n, np.int64); src[rs] = src_sorted; src = src[rows]
        key = c[(src + hh) % n]
        looked.append(len(rows))
        o2 = np.lexsort((key, g))                                # inside every group by key (stable)
        r2, g2, k2 = rows[o2], g[o2], key[o2]
        newhead = np.r_[True, (g2[1:] != g2[:-1]) | (k2[1:] != k2[:-1])]
        gstart = np.r_[True, g2[1:] != g2[:-1]]
        base = np.maximum.accumulate(np.where(gstart, np.arange(len(r2)), 0))
        sub = np.maximum.accumulate(np.where(newhead, np.arange(len(r2)), 0)) - base
        newc = g2 + sub                                          # the class of a subgroup: the group's first index + its offset in the group
        order[g2 + (np.arange(len(r2)) - base)] = r2
        c2 = c.copy(); c2[r2] = newc
        c = c2
        hh *= 2; rounds += 1
    return order, rounds, looked


t0 = time.time()
plain, r0, l0 = run(False)
t1 = time.time()
chained, r1, l1 = run(True)
t2 = time.time()
print("block of %d bytes at %d: %d rows unsorted after the round for %d bytes" % (n, off, l0[0] if l0 else 0, h))
print("plain rounds  : %2d, rows looked at per round %s (sum %d; %.0f s)" % (r0, l0, sum(l0), t1 - t0))
print("chained rounds: %2d, rows looked at per round %s (sum %d; %.0f s)" % (r1, l1, sum(l1), t2 - t1))
# the same order up to rows whose rotations are equal (periodic blocks only)
same = np.array_equal(plain, chained)
print("same order:", same)
if not same:
    d = np.nonzero(plain != chained)[0]
    print("  first difference at sorted index", d[0], plain[d[0]], chained[d[0]])
