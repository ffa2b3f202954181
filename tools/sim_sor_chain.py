"""Functional model of sor_chain.hip on the CPU (test infrastructure for the kernel's INDEX ARITHMETIC and WAIT THRESHOLDS; not a product path).

Every formula below is transcribed from the kernel (same names).  Visibility is modelled adversarially:
  * a global store becomes visible only when the progress word that covers it is published (the latest the protocol allows);
  * a global load reads memory when it is issued (the earliest the protocol allows);
  * per round every workgroup advances by at most one interval, consumers first (every consumer runs as early as the thresholds permit);
  * LDS: within one barrier interval no wave may read an address another wave writes (checked).
The result must equal the raster-order oracle bit for bit.

usage: sim_sor_chain.py [W H K FA NA FB NB_ [nb]]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

f32 = np.float32
CH, AH = 4, 2
PUBD = int(os.environ.get("SIM_PUBD", "2"))     # intervals between a store and the progress word that covers it: 2, or 1 for the shapes launched with <..., PL = 1, PUBD = 1>
LEAD = AH + 1
OOB = None
# visibility model: "raw" = stores land as late, loads sample as early as the protocol allows (read-after-write hazards);
# "war" = stores land when issued, loads sample when their data is consumed, AH intervals later (write-after-read hazards on the in-place x plane)
MODE = os.environ.get("SIM_MODE", "raw")
SLACK = int(os.environ.get("SIM_SLACK", "0"))      # > 0: weaken every wait threshold by that many intervals (the model must then FAIL)


def start_shift(S):
    """sor_chain.hip chain_start_shift / ChainLds::OPRING: chunks by which the groups of a shape with an operand ring start early"""
    return (2 if S.KG <= 10 else 4) if ring_rows(S)[0] > 0 else 0


def round_up(a, m):
    return (a + m - 1) // m * m


def ring_rows(S):
    """(ChainLds::OPR, the derived minimum) of a shape: rows of the LDS operand ring (0 = the shape has none)"""
    NW, KG, FMAX = S.NW, S.KG, S.FMAX
    ring = 2 * CH * 64 * 8
    tv0 = (NW + 1) * ring; tvw = 2 * CH * (FMAX + 1) * 8; es0 = tv0 + NW * tvw; esw = 2 * (FMAX - 1 if FMAX > 1 else 1) * CH * 8
    ops0 = (es0 + NW * esw + 64 * 8 + esw + 15) & ~15
    oprowb = (64 + KG - 1) * 16
    derived = 3 * NW + 2 * KG + 2                                  # the smallest depth without a write-after-read clash at one step of read-ahead (ring_hazards)
    oprmin = 3 * (NW - 1) + 2 * (KG - 1) + 9                       # ChainLds::OPRMIN = derived + 2: the kernel keeps two rows of margin, three where they fit
    fits = lambda r: ops0 + 2 * r * oprowb + 32 <= 160 * 1024
    pfl = ring_prefetch(S)[1]
    tight = derived - (pfl - 1 if isinstance(pfl, int) else 0)     # ChainLds::OPRTIGHT: the bound itself, lowered by the later stages' read-ahead (3,2,2,2,2,2,2: 51 rows)
    opr = oprmin + 1 if fits(oprmin + 1) else oprmin if fits(oprmin) else tight
    if ring_infill(S): opr = (oprmin + 1 + 4 * AH + 3) // 4 * 4   # ChainLds::OPRFILL: the FILL wave's groups of four rows are in flight AH intervals earlier than the IN wave wrote
    return (opr, derived) if KG <= 16 and fits(opr) else (0, derived)


def ring_prefetch(S):
    """ChainShape::PF0 / PFL: steps by which the ring-fed sweeps of the first stage / of the later stages read their rows ahead -- (2, 3) where every stage has at least
    two sweeps, "chunk" (the rows of steps 4 c + 1 .. 4 c + 4 at the top of chunk c) where every stage has one; 1 otherwise and for workgroups of 8 stages or more"""
    if S.NW >= 8: return (1, 1)
    if S.FA >= 2 and (S.NB_ == 0 or S.FB >= 2): return (2, 3)
    if S.FA == 1 and S.NB_ == 0: return ("chunk", "chunk")
    return (1, 1)


def ring_infill(S):
    """ChainShape::INFILL: a FILL wave copies whole operand rows (the entries above the band and the band's own) into the ring by LDS-DMA and the first stage is ring-fed
    like the others (the one-sweep shapes)"""
    return S.FA == 1 and S.NB_ == 0 and S.NW < 8


def ring_hazards(S, OPR, PF, nsteps=600, infill=False):
    """The LDS operand ring of chain_compute under the barrier lockstep (compute wave w runs its chunk c in interval LEAD + w + c; accesses of DIFFERENT waves inside one
    interval are unordered, a wave's own accesses keep program order).  Row r (the first stage's step r) is written by the IN wave (the KG - 1 entries above the band) in
    interval r // CH + AH and by the first stage in its step r (interval LEAD + r // CH); sweep kappa of stage w uses row i + w - 2 kappa at its step i and reads it at
    the END of step i - PF (before the first barrier when i < PF); row r + OPR lands in row r's slot.  Returns the violated dependencies.
    infill (ChainShape::INFILL): nobody but the FILL wave writes the ring -- the DMA of rows 4 c .. 4 c + 3 is issued in interval c and has landed when interval c + 1
    ends (chain_fill's counted wait): a row may change at any time in between --, and the first stage's sweep reads the ring like the others'."""
    bad = []
    w_in = lambda r: r // CH + AH
    f_issue = lambda r: r // CH                                     # infill: the interval in which the DMA of row r is issued ...
    f_done = lambda r: r // CH + 1                                  # ... and the one by whose end it has landed
    w_st = lambda r: LEAD + r // CH
    PF_of = PF if isinstance(PF, tuple) else (PF, PF)              # (first stage, later stages)
    for w in range(S.NW):
        PF = PF_of[0] if w == 0 else PF_of[1]
        for f in range(S.Fw(w)):
            kap = S.kw(w) + f
            if w == 0 and f == 0 and not infill: continue                    # the group's first sweep loads from memory
            for i in range(nsteps):
                rho = i + w - 2 * kap
                if PF == "chunk":                                            # read at the top of the chunk of step i - 1 (step 0: before the first barrier); ring-fed stages w >= 1 only
                    if w == 0 and not infill: bad.append(("raw-self", w, kap, i)); continue
                    issue = -1 if i == 0 else (i - 1) // CH * CH - 1         # "behind step issue": the chunk top lies behind the last step of the chunk before
                    t_read = 0 if i == 0 else LEAD + w + (i - 1) // CH
                else:
                    issue = i - PF
                    t_read = 0 if issue < 0 else LEAD + w + issue // CH
                if infill:
                    if rho >= 0 and t_read > 0 and not f_done(rho) < t_read: bad.append(("raw-fill", w, kap, i))   # (reads before the first barrier see the initial zeros: steps in the guards)
                    if rho + OPR >= 0 and not t_read < f_issue(rho + OPR): bad.append(("war", w, kap, i))
                    continue
                if rho >= 0:                                                 # read after write (rows < 0 are never written: the ring's initial zeros)
                    if w == 0 and not rho <= issue: bad.append(("raw-self", w, kap, i))          # the first stage's own trailing sweeps: program order inside the wave
                    if w > 0 and not w_st(rho) < t_read: bad.append(("raw", w, kap, i))
                    if not w_in(rho) < t_read: bad.append(("raw-in", w, kap, i))
                if rho + OPR >= 0 and not t_read < w_in(rho + OPR): bad.append(("war", w, kap, i))   # the slot's next row arrives (IN wave first) after this read
    return bad


class Shape:
    def __init__(self, FA, NA, FB, NB_):
        self.FA, self.NA, self.FB, self.NB_ = FA, NA, FB, NB_
        self.NW = NA + NB_
        self.KG = NA * FA + NB_ * FB
        self.FMAX = max(FA, FB if NB_ else FA)

    def Fw(self, w):
        return self.FA if w < self.NA else self.FB

    def kw(self, w):
        return w * self.FA if w < self.NA else self.NA * self.FA + (w - self.NA) * self.FB


def prepare(sysm, W, H, K, S, nb):
    """SorWorkspace::configure (chain branch) + k_sor_prepare"""
    ws = type("WS", (), {})()
    ws.NB = (H + K - 1 + 63) // 64
    ws.NG = K // S.KG
    ws.G = K + 64
    ws.RP = round_up(H + 2 * ws.G, 16)
    ws.NCH = round_up((W + 64 + S.KG - S.NW + 2 * CH + S.FMAX + CH - 1) // CH + start_shift(S), 4)
    ws.NS = round_up(ws.NCH + LEAD + S.NW + 2, AH)
    ws.ND = ws.NS * CH + 64 * ws.NB + ws.G + 32
    ws.ent = ws.ND * ws.RP
    ws.EP = 256
    ws.Wp = round_up(ws.EP + ws.NS * CH + 64 + 2 * K + 64, 8)
    ws.edge_job = ws.NB * K * ws.Wp
    sa = np.zeros((nb, ws.ent, 4), f32); sb = np.zeros((nb, ws.ent, 4), f32); x = np.zeros((nb, ws.ent, 2), f32)
    for job in range(nb):
        s = sysm[job]
        for r in range(H):
            for c in range(W):
                hp = s["sh"][r, c]; hl = s["sh"][r, c - 1] if c > 0 else f32(0)
                vp = s["sv"][r, c]; vt = s["sv"][r - 1, c] if r > 0 else f32(0)
                d = f32(hl + hp)
                if r > 0: d = f32(d + vt)
                if r < H - 1: d = f32(d + vp)
                m12 = s["a12"][r, c]
                A11 = f32(s["a22"][r, c] + d); A22 = f32(s["a11"][r, c] + d)
                det = f32(f32(A11 * A22) - f32(m12 * m12))
                e = (c + r + ws.G) * ws.RP + (r + ws.G)
                sa[job, e] = (f32(A11 / det), f32(m12 / f32(-det)), f32(A22 / det), vt)
                sb[job, e] = (s["b1"][r, c], s["b2"][r, c], hp, vp if r < H - 1 else f32(0))
                x[job, e] = (s["du"][r, c], s["dv"][r, c])
    ws.sa, ws.sb, ws.x = sa, sb, x
    ws.edge = np.zeros((nb * ws.edge_job, 2), f32)
    ws.flags = np.zeros(nb * ws.NB * ws.NG, np.int64)
    ws.order = [(b, key - 3 * b) for key in range(3 * (ws.NB - 1) + ws.NG) for b in range(ws.NB) if 0 <= key - 3 * b < ws.NG]
    return ws


def sor_point2(selfv, right, top, bottom, left, hlz, SA, SB, omega):
    """(n,2) float32 arrays; SA, SB (n,4)"""
    s = SB[:, 2:3] * right
    s = s + SA[:, 3:4] * top
    s = s + SB[:, 3:4] * bottom
    s = s + SB[:, 0:2]
    B = hlz[:, 0:1] * left + s
    t2 = np.stack([SA[:, 1] * B[:, 1], SA[:, 2] * B[:, 1]], 1)
    t = SA[:, 0:2] * B[:, 0:1] + t2
    return (selfv + f32(omega) * (t - selfv)).astype(f32)


def shr1(v, fill):
    """lane l <- lane l-1, lane 0 <- fill (a (2,) value)"""
    o = np.empty_like(v)
    o[1:] = v[:-1]
    o[0] = fill
    return o


class WG:
    """one workgroup = (job, b, g): NW compute waves + the I/O wave, advanced one barrier interval at a time"""

    def __init__(self, ws, S, W, H, K, omega, job, b, g, nb):
        self.ws, self.S, self.W, self.H, self.K, self.omega, self.job, self.b, self.g, self.nb = ws, S, W, H, K, f32(omega), job, b, g, nb
        NW = S.NW
        self.c_first = (((g * (S.KG - NW)) // CH) & ~1) - start_shift(S)
        self.c_first_prev = ((((g - 1) * (S.KG - NW)) // CH) & ~1) - start_shift(S) if g > 0 else 0
        self.I = 0
        self.done = False
        # LDS
        self.ring = np.zeros((NW + 1, 2, CH, 64, 2), f32)
        self.tv = [np.zeros((2, CH, S.Fw(w) + 1, 2), f32) for w in range(NW)]
        self.es = [np.zeros((2, max(S.Fw(w) - 1, 1), CH, 2), f32) for w in range(NW)]
        # compute waves' state
        self.cw = []
        for w in range(NW):
            F = S.Fw(w)
            st = g * NW + w; k0 = g * S.KG + S.kw(w); O = k0 - st + 1
            d = dict(F=F, k0=k0, s_start=self.c_first * CH - O, r0=64 * b - k0, res=np.zeros((F, 64, 2), f32), selfv=np.zeros((F, 64, 2), f32),
                     hlz=np.zeros((F, 64, 2), f32), step=0)
            assert d["s_start"] <= -1
            self.cw.append(d)
        # I/O wave state
        self.io_init()

    # ---- compute wave w, one chunk (parity par) -----------------------------------------------------------------------------------
    def compute_chunk(self, w, par, reads, writes):
        ws, d = self.ws, self.cw[w]
        F, RP = d["F"], ws.RP
        FOFF = 2 * RP + 1
        lane = np.arange(64)
        E0 = (d["r0"] + ws.G) * RP + (d["r0"] + ws.G) + d["s_start"] * RP - (F - 1) * FOFF
        for j in range(CH):
            reads.add(("ring", w, par, j)); reads.add(("tv", w, par, j))
            bottom0 = self.ring[w, par, j].copy()
            fl = self.tv[w][par, j]
            right0 = shr1(bottom0, fl[0])
            sh = [shr1(d["res"][f], fl[f + 1]) for f in range(F)]
            nres = []
            for f in range(F):
                right = right0 if f == 0 else sh[f - 1]
                bottom = bottom0 if f == 0 else d["res"][f - 1]
                ent = E0 + d["step"] * RP + lane + (F - 1 - f) * FOFF          # vo[f] / 16 + so / 16
                assert ent.min() >= 0 and ent.max() < ws.ent, "operand load outside the planes"
                SA = ws.sa[self.job, ent]; SB = ws.sb[self.job, ent]
                nres.append(sor_point2(d["selfv"][f], right, sh[f], bottom, d["res"][f], d["hlz"][f], SA, SB, self.omega))
                d["selfv"][f] = right
                d["hlz"][f] = SB[:, 2:4]
            for f in range(F):
                d["res"][f] = nres[f]
            for f in range(F - 1):
                self.es[w][par, f, j] = d["res"][f][63]; writes.add(("es", w, par, f, j))
            self.ring[w + 1, par, j] = d["res"][F - 1]; writes.add(("ring", w + 1, par, j))
            d["step"] += 1

    # ---- I/O wave -------------------------------------------------------------------------------------------------------------------
    def io_init(self):
        ws, S, b, g, job = self.ws, self.S, self.b, self.g, self.job
        NW, K, RP = S.NW, self.K, ws.RP
        self.has_up, self.has_prev = b > 0, g > 0
        st0, k0g = g * NW, g * S.KG
        self.st0 = st0
        O0 = k0g - st0 + 1; self.s_start0 = self.c_first * CH - O0; r00 = 64 * b - k0g
        U00 = (r00 + ws.G) * RP + (r00 + ws.G)
        self.Xin_base = U00 + self.s_start0 * RP
        Fl = S.Fw(NW - 1); self.Fl = Fl
        k0l = k0g + S.kw(NW - 1); Ol = k0l - (st0 + NW - 1) + 1; self.s_startl = self.c_first * CH - Ol; r0l = 64 * b - k0l
        U0l = (r0l + ws.G) * RP + (r0l + ws.G)
        FOFF = 2 * RP + 1
        self.Xout_base = U0l + self.s_startl * RP - (Fl - 1) * FOFF
        self.r0l = r0l
        # (1) tv lanes
        self.tvl = []
        for w in range(NW):
            for fi in range(S.Fw(w) + 1):
                for j in range(CH):
                    k0w = k0g + S.kw(w); Ow = k0w - (st0 + w) + 1
                    s0 = (self.c_first - w) * CH - Ow
                    row = k0w - 1 + fi; col = s0 + j + (1 if fi == 0 else -(fi - 1))
                    tvx = st0 + w == 0 and fi == 0
                    off = job * ws.edge_job + ((b - 1) * K + row) * ws.Wp + ws.EP + col
                    voff = off if (self.has_up and not tvx) else OOB
                    tvx_voff = (s0 - self.s_start0 + j + 1) * RP if tvx else None
                    if voff is not None: assert off >= 0
                    self.tvl.append(dict(w=w, fi=fi, j=j, voff=voff, tvx=tvx, tvx_voff=tvx_voff))
        # (2) edge-store lanes
        self.esl = []
        for w in range(NW):
            for f in range(S.Fw(w)):
                for j in range(CH):
                    k0w = k0g + S.kw(w); Ow = k0w - (st0 + w) + 1
                    s0 = (self.c_first - 1 - LEAD - w) * CH - Ow
                    off = job * ws.edge_job + (b * K + k0w + f) * ws.Wp + ws.EP + (s0 + j - 63 - f)
                    assert off >= 0
                    self.esl.append(dict(w=w, f=f, j=j, voff=off, lo=LEAD + w + 1))
        D63 = 63 // CH
        dcf = self.c_first - self.c_first_prev
        self.need_up0 = LEAD + 3 + D63
        self.need_up20 = 1 + D63 + dcf + LEAD + NW - 1 + 2
        self.need_prev0 = dcf + LEAD + NW - 1 + 2
        self.cap = ws.NCH + LEAD + NW                      # every real chunk of a producer is published at this count
        self.tvr = [[None] * len(self.tvl) for _ in range(AH)]
        self.xr = [[np.zeros((64, 2), f32) for _ in range(CH)] for _ in range(AH)]
        for q in range(AH):
            self.tvr[q] = [np.zeros(2, f32) for _ in self.tvl]
        self.so_tv = self.so_es = self.so_xin = self.so_xout = 0
        self.s_out = self.s_startl - (LEAD + NW) * CH
        self.pending_stores = []       # stores of S(I): become visible when the flag that covers them is published
        self.flag_idx = (job * ws.NB + b) * ws.NG + g

    def io_blocked(self):
        """the waits of part (d) of interval I: True if one of them is not satisfied yet"""
        ws, I = self.ws, self.I
        fl = ws.flags
        if self.has_up and fl[self.flag_idx - ws.NG] < min(self.need_up0 + I - SLACK, self.cap): return True
        if self.has_up and self.has_prev and fl[self.flag_idx - ws.NG - 1] < min(self.need_up20 + I - SLACK, self.cap): return True
        if self.has_prev and fl[self.flag_idx - 1] < min(self.need_prev0 + I - SLACK, self.cap): return True
        return False

    def io_interval_front(self, reads, writes):
        """(a) (b) (c) of interval I -- everything in front of the waits"""
        ws, S, I = self.ws, self.S, self.I
        NW, RP, W, H = S.NW, ws.RP, self.W, self.H
        q, p = I % AH, I & 1
        # (a)
        def sample(v):          # "war": the load is sampled now, not when it was issued
            if isinstance(v, tuple):
                arr, idx = v
                return arr[idx].copy()
            return v
        for n, L in enumerate(self.tvl):
            cp = (p + L["w"]) & 1
            self.tv[L["w"]][cp, L["j"], L["fi"]] = sample(self.tvr[q][n]); writes.add(("tv", L["w"], cp, L["j"]))
        for j in range(CH):
            self.ring[0, p, j] = sample(self.xr[q][j]); writes.add(("ring", 0, p, j))
        # (b) stores of this interval (visible later)
        new = []
        for L in self.esl:
            w, f, j = L["w"], L["f"], L["j"]
            cp = (p + w) & 1
            if f < S.Fw(w) - 1:
                v = self.es[w][cp, f, j].copy(); reads.add(("es", w, cp, f, j))
            else:
                v = self.ring[w + 1, cp, j, 63].copy(); reads.add(("ring", w + 1, cp, j))
            if L["lo"] <= I < L["lo"] + ws.NCH:
                new.append(("edge", L["voff"] + self.so_es, v))
        act = LEAD + NW <= I < LEAD + NW + ws.NCH
        po = (p + NW - 1) & 1
        lane = np.arange(64)
        rl = self.r0l + lane - (self.Fl - 1)
        row_ok = (rl >= 0) & (rl < H)
        for j in range(CH):
            v = self.ring[NW, po, j].copy(); reads.add(("ring", NW, po, j))
            col = self.s_out + j - (lane + self.Fl - 1)
            ok = act & row_ok & (col >= 0) & (col < W)
            ent = self.Xout_base + self.so_xout + j * RP + lane
            for l in np.nonzero(ok)[0]:
                new.append(("x", ent[l], v[l]))
        # (c) publish (OUT wave): the stores of interval I - PUBD are complete
        if MODE == "war":
            for kind, addr, v in new: (ws.edge if kind == "edge" else ws.x[self.job])[addr] = v
            new = []
        self.pending_stores.append(new)
        if len(self.pending_stores) > PUBD:
            for kind, addr, v in self.pending_stores.pop(0):
                (ws.edge if kind == "edge" else ws.x[self.job])[addr] = v
        ws.flags[self.flag_idx] = max(I - PUBD + 1, 0)

    def io_interval_back(self):
        """(e): the loads of interval I, issued once the waits are satisfied"""
        ws, I = self.ws, self.I
        RP = ws.RP
        q = I % AH
        for n, L in enumerate(self.tvl):
            late = MODE == "war"
            if L["tvx"]:
                e = self.Xin_base + L["tvx_voff"] + self.so_xin
                self.tvr[q][n] = (ws.x[self.job], e) if late else ws.x[self.job, e].copy()
            elif L["voff"] is None:
                self.tvr[q][n] = np.zeros(2, f32)
            else:
                self.tvr[q][n] = (ws.edge, L["voff"] + self.so_tv) if late else ws.edge[L["voff"] + self.so_tv].copy()
        lane = np.arange(64)
        for j in range(CH):
            e = self.Xin_base + self.so_xin + (j + 1) * RP + lane + 1          # (c, r + 1): one diagonal further
            assert e.max() < ws.ent, "x load outside the plane"
            self.xr[q][j] = (ws.x[self.job], e) if MODE == "war" else ws.x[self.job, e].copy()
        self.so_tv += CH; self.so_es += CH; self.so_xin += CH * RP; self.s_out += CH
        if I >= LEAD + self.S.NW: self.so_xout += CH * RP

    # ---- one barrier interval; returns False if the I/O wave is blocked (nothing done) -----------------------------------------------
    def advance(self):
        ws, S = self.ws, self.S
        if self.done: return False
        if not getattr(self, "_front_done", False):
            reads, writes = set(), set()
            # all waves of the interval: compute waves first or last makes no difference if the read / write sets are disjoint
            for w in range(S.NW):
                c = self.I - LEAD - w
                if 0 <= c < ws.NCH: self.compute_chunk(w, c & 1, reads, writes)
            self.io_interval_front(reads, writes)
            clash = {r for r in reads if r in writes}
            assert not clash, f"LDS read/write clash within interval {self.I}: {sorted(clash)[:4]}"
            self._front_done = True
        if self.io_blocked(): return False
        self.io_interval_back()
        self._front_done = False
        self.I += 1
        if self.I == ws.NS:
            for st in self.pending_stores:
                for kind, addr, v in st:
                    (ws.edge if kind == "edge" else ws.x[self.job])[addr] = v
            ws.flags[self.flag_idx] = 0x7fffffff
            self.done = True
        return True


def run(W, H, K, S, nb=1, seed=0, omega=1.9):
    from synth import copy_sys, sor_system
    import oracle as orc
    rng = np.random.default_rng(seed)
    systems = [sor_system(rng, W, H) for _ in range(nb)]
    for s in systems:
        s["du"][:, :W] = rng.uniform(-.2, .2, (H, W)); s["dv"][:, :W] = rng.uniform(-.2, .2, (H, W))
    ws = prepare(systems, W, H, K, S, nb)
    wgs = [WG(ws, S, W, H, K, omega, t % nb, *ws.order[t // nb], nb) for t in range(nb * ws.NB * ws.NG)]
    # consumers first: per round every workgroup advances by at most ONE interval, in reverse ticket order, so that a consumer always
    # runs the moment its thresholds let it (a producer is never further ahead than the protocol demands)
    while not all(g.done for g in wgs):
        progressed = False
        for g in reversed(wgs):
            if g.advance(): progressed = True
        assert progressed, "deadlock: " + str([(g.b, g.g, g.I) for g in wgs if not g.done])
    o = orc.Oracle()
    worst = 0
    for job, s in enumerate(systems):
        a = copy_sys(s)
        o.sor(a["du"], a["dv"], a["a11"], a["a12"], a["a22"], a["b1"], a["b2"], a["sh"], a["sv"], W, K, omega)
        bad = 0
        for r in range(H):
            for c in range(W):
                v = ws.x[job, (c + r + ws.G) * ws.RP + r + ws.G]
                if not (v[0] == a["du"][r, c] and v[1] == a["dv"][r, c]): bad += 1
        worst = max(worst, bad)
    return worst, ws


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    W, H, K, FA, NA, FB, NB_ = a[:7] if len(a) >= 7 else (40, 70, 6, 1, 3, 1, 0)
    nb = a[7] if len(a) > 7 else 1
    bad, ws = run(W, H, K, Shape(FA, NA, FB, NB_), nb)
    print(f"W={W} H={H} K={K} shape=({FA}x{NA},{FB}x{NB_}) NB={ws.NB} NG={ws.NG} NCH={ws.NCH} NS={ws.NS}: mismatching points = {bad}")
    sys.exit(1 if bad else 0)
