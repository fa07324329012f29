"""Discrete check of the DMA / barrier schedules of tools/micro/gemm256_stagger.hip and gemm256_strip.hip in the model their
headers argue in: barrier intervals.  Group 0 (wave row 0) runs the read section of phase ph in interval 2 ph and its multiply
section in 2 ph + 1; group 1 one interval later; barrier k ends interval k.  Every DMA item is two instructions per wave; a
wave's `s_waitcnt vmcnt(N)` at the end of a multiply section retires all but its N youngest instructions, in order.

  RAW  an item may be read in interval k only if BOTH groups' retiring waits lie in intervals < k (wait, then a barrier);
  WAR  an item may be issued into an LDS region in interval k only if every earlier read of the region was issued in an
       interval <= k - 2 (the reader waits lgkmcnt(0) at the start of its next interval; a barrier must lie between);
  END  everything is retired before the epilogue reuses the ring.

The schedules below are transcribed from the kernels' phase lambdas (same expressions).  usage: python tools/micro/schedule_check.py
"""
import sys


class Sched:
    def __init__(self, name):
        self.name = name
        self.pro_issues, self.pro_wait, self.pro_reads = [], 0, []
        self.phases = []          # dicts: reads [(item)], issues [(item, region)], wait N
        self.final_drain = True

    def check(self):
        errs = []
        region_of = {}
        # per group timelines
        retire = [{}, {}]
        for g in (0, 1):
            issued = []           # (item, interval) in issue order
            def do_wait(N, interval):
                keep = N // 2     # items allowed in flight
                done = issued[:len(issued) - keep] if keep else issued[:]
                for it, _ in done:
                    retire[g].setdefault(it, interval)
            for it, reg in self.pro_issues:
                issued.append((it, -1))
            do_wait(self.pro_wait, -1)
            for ph, P in enumerate(self.phases):
                for it, reg in P["issues"]:
                    issued.append((it, 2 * ph + g))
                do_wait(P["wait"], 2 * ph + 1 + g)
            end = 2 * len(self.phases) + 2
            if self.final_drain:
                do_wait(0, end)
            for it, _ in issued:
                if it not in retire[g]:
                    errs.append(f"END: {it} never retired by group {g}")
        visible = {}
        for it in retire[0]:
            visible[it] = max(retire[0][it], retire[1].get(it, 10 ** 9)) + 1
        # reads
        reads = []                # (item, group, interval)
        for it in self.pro_reads:
            for g in (0, 1):
                reads.append((it, g, 0))
        for ph, P in enumerate(self.phases):
            for it in P["reads"]:
                for g in (0, 1):
                    reads.append((it, g, 2 * ph + g))
        for it, g, k in reads:
            if it.startswith("dummy"):
                continue
            if it not in visible:
                errs.append(f"RAW: {it} read in interval {k} but never issued")
            elif k < visible[it]:
                errs.append(f"RAW: {it} read by group {g} in interval {k}, visible from {visible[it]}")
        # WAR: issues in time order per region
        events = []               # (interval, kind, item, region)
        for it, reg in self.pro_issues:
            events.append((-1, "issue", it, reg)); region_of[it] = reg
        for ph, P in enumerate(self.phases):
            for it, reg in P["issues"]:
                region_of[it] = reg
                for g in (0, 1):
                    events.append((2 * ph + g, "issue", it, reg))
        last_read = {}            # region -> latest read interval of its current content
        content = {}
        # process in chronological order; reads of an item belong to its region
        rd_by_item = {}
        for it, g, k in reads:
            rd_by_item.setdefault(it, []).append(k)
        for k, kind, it, reg in sorted(events, key=lambda e: e[0]):
            if reg is None:
                continue
            prev = content.get(reg)
            if prev is not None and prev != it:
                for kr in rd_by_item.get(prev, []):
                    if k < kr + 2:
                        errs.append(f"WAR: {it} issued into region {reg} in interval {k}, {prev} still read in interval {kr}")
                if prev not in rd_by_item and not prev.startswith("dummy"):
                    errs.append(f"UNUSED: {prev} overwritten by {it} without ever being read")
            content[reg] = it
        return errs


def gemm256(KT, D):
    """gemm256<..., D>: stream g = 4 t + kind (B0, A0, B1, A1), slot g & 7; phase ph = 4 t + q issues g = ph + D."""
    s = Sched(f"gemm256 D={D} KT={KT}")
    G = 4 * KT
    kind = ["B0", "A0", "B1", "A1"]
    item = lambda g: f"{kind[g & 3]}({g >> 2})"
    npro = 5 if D == 5 else 7
    s.pro_issues = [(item(g), g & 7) for g in range(npro)]
    s.pro_wait = 4 if D == 5 else 8
    s.pro_reads = [item(0)]
    for t in range(KT):
        for q in range(4):
            ph = 4 * t + q
            reads = {0: [item(4 * t + 1)], 1: [item(4 * t + 2)], 2: [item(4 * t + 3)], 3: ([item(4 * (t + 1))] if t + 1 < KT else [])}[q]
            issues = [(item(ph + D), (ph + D) & 7)] if ph + D < G else []
            if D == 7 and ph + 7 < G: w = 8
            elif D == 7 and ph + 6 < G: w = 6
            elif ph + 5 < G: w = 4
            elif ph + 4 < G: w = 2
            else: w = 0
            s.phases.append({"reads": reads, "issues": issues, "wait": w})
    s.final_drain = False        # the last phases' own waits must have drained everything
    return s


def gemm256x128(KT):
    """gemm256x128: kinds B, A0, A1; slot (3 t + kind) % 9; p0(t) issues A0(t+2); p1(t) issues A1(t+2), B(t+3)."""
    s = Sched(f"gemm256x128 KT={KT}")
    B = lambda t: (f"B({t})", (3 * t) % 9)
    A0 = lambda t: (f"A0({t})", (3 * t + 1) % 9)
    A1 = lambda t: (f"A1({t})", (3 * t + 2) % 9)
    s.pro_issues = [B(0), A0(0), A1(0), B(1), A0(1), A1(1), B(2)]
    s.pro_wait = 6
    s.pro_reads = ["B(0)"]
    for t in range(KT):
        s.phases.append({"reads": [f"A0({t})"], "issues": [A0(t + 2)] if t + 2 < KT else [], "wait": 6 if t + 2 < KT else 0})
        iss = ([A1(t + 2)] if t + 2 < KT else []) + ([B(t + 3)] if t + 3 < KT else [])
        s.phases.append({"reads": [f"A1({t})"] + ([f"B({t + 1})"] if t + 1 < KT else []), "issues": iss, "wait": 6 if t + 3 < KT else 0})
    s.final_drain = False
    return s


def strip(KT):
    """gemm256_strip: every phase issues one two-instruction item; vmcnt(8) after odd phases, vmcnt(10) after even ones."""
    s = Sched(f"gemm256_strip KT={KT}")
    NG = KT // 3
    cnt = [0]
    def dummy():
        cnt[0] += 1
        return (f"dummy{cnt[0]}", None)
    def Bh(h, T):
        return (f"B{h}({T})", f"ring{(2 * T + h) & 3}") if T < KT else dummy()
    def S(piece, G):
        return (f"S{piece}({G})", f"strip{G & 1}.{piece}") if G < NG else dummy()
    s.pro_issues = [S(0, 0), dummy(), S(1, 0), dummy(), S(2, 0), Bh(0, 0), dummy(), Bh(1, 0), dummy(), Bh(0, 1), dummy(), Bh(1, 1)]
    s.pro_wait = 8
    s.pro_reads = ["B0(0)"]
    for T in range(KT):
        Gc, dxi = T // 3, T % 3
        for q in range(4):
            if q == 0: reads = [f"S0({Gc})", f"S1({Gc})", f"S2({Gc})"]      # A0 rows of the strip (any piece may be touched: rows i + dx - 1, 256-258)
            elif q == 1: reads = [f"B1({T})"]
            elif q == 2: reads = [f"S0({Gc})", f"S1({Gc})", f"S2({Gc})"]
            else: reads = [f"B0({T + 1})"] if T + 1 < KT else []
            if q == 1: iss = [Bh(0, T + 2)]
            elif q == 3: iss = [Bh(1, T + 2)]
            else:
                piece = 2 * dxi + (q >> 1)
                iss = [S(piece, Gc + 1)] if piece < 3 else [dummy()]
            ph = 4 * T + q
            s.phases.append({"reads": reads, "issues": iss, "wait": 8 if ph & 1 else 10})
    s.final_drain = True
    return s


if __name__ == "__main__":
    bad = 0
    for sc in [gemm256(8, 5), gemm256(36, 5), gemm256(2, 5), gemm256(8, 7), gemm256(36, 7), gemm256(2, 7), gemm256(6, 7),
               gemm256x128(3), gemm256x128(9), gemm256x128(18), gemm256x128(36), strip(6), strip(18), strip(36)]:
        errs = sc.check()
        print(f"{sc.name:28s} {'ok' if not errs else 'VIOLATIONS: %d' % len(errs)}")
        for e in errs[:6]:
            print("   ", e)
        bad += bool(errs)
    sys.exit(1 if bad else 0)
