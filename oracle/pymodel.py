"""Independent big-integer model of the hot path (TEST INFRASTRUCTURE ONLY).

Second, *mathematical* statement of what the reference computes, written with Python
integers and textbook definitions (NTT as a polynomial evaluation, ModDown as an exact
floor division after CRT reconstruction, negacyclic schoolbook products).  It shares no
code with oracle/*.c; the C oracle is checked against it (tests/test_oracle_vs_model.py)
and it generates the committed fixtures under tests/golden/ (tests/golden/make_golden.py).

Conventions follow lattigo v2.3.0 as used by SNUCP/MKHE-KKLSS:
  * Montgomery radix R = 2^64,
  * forward NTT: natural order in, bit-reversed order out,
      NTT(a)[j] = sum_i a_i * psi^((2*bitrev(j)+1) * i)  mod q,
  * psi = g^((q-1)/2N) with g the first primitive root found scanning g = 3, 4, ...
PARITY UNPINNED vs Go: there are no reference golden vectors.
"""
import math
import random

R = 1 << 64


# --------------------------------------------------------------------------- number theory
def _is_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _rho(n, rng):
    if n % 2 == 0:
        return 2
    while True:
        c = rng.randrange(1, n)
        x = y = rng.randrange(0, n)
        d = 1
        while d == 1:
            x = (x * x + c) % n
            y = (y * y + c) % n
            y = (y * y + c) % n
            d = math.gcd(abs(x - y), n)
        if d != n:
            return d


def prime_factors(n):
    rng = random.Random(12345)
    out = set()
    stack = [n]
    while stack:
        m = stack.pop()
        if m == 1:
            continue
        if _is_prime(m):
            out.add(m)
            continue
        d = _rho(m, rng)
        stack += [d, m // d]
    return sorted(out)


def primitive_root(q):
    """lattigo ring/primes.go primitiveRoot: g = 2; loop { g++ ; test }."""
    fs = prime_factors(q - 1)
    g = 2
    while True:
        g += 1
        if all(pow(g, (q - 1) // f, q) != 1 for f in fs):
            return g


def find_psi(q, N):
    return pow(primitive_root(q), (q - 1) // (2 * N), q)


def bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


# --------------------------------------------------------------------------- literal scalar ops
def mred(x, y, q, qinv):
    m = x * y
    mhi, mlo = m >> 64, m & (R - 1)
    hhi = (((mlo * qinv) & (R - 1)) * q) >> 64
    r = (mhi - hhi + q) & (R - 1)
    return r - q if r >= q else r


def qinv64(q):
    return pow(q, -1, R)


def mform(a, q):
    return (a << 64) % q


def invmform(a, q):
    return a * pow(R, -1, q) % q


# --------------------------------------------------------------------------- transforms by definition
def ntt_def(a, q, psi, logN):
    N = 1 << logN
    out = [0] * N
    for j in range(N):
        w = pow(psi, 2 * bitrev(j, logN) + 1, q)
        acc, wp = 0, 1
        for i in range(N):
            acc += a[i] * wp
            wp = wp * w % q
        out[j] = acc % q
    return out


def intt_def(A, q, psi, logN):
    N = 1 << logN
    ninv = pow(N, -1, q)
    out = [0] * N
    ws = [pow(psi, -(2 * bitrev(j, logN) + 1), q) for j in range(N)]
    for i in range(N):
        acc = 0
        for j in range(N):
            acc += A[j] * pow(ws[j], i, q)
        out[i] = acc * ninv % q
    return out


def negacyclic_mul(a, b, q):
    N = len(a)
    out = [0] * N
    for i, ai in enumerate(a):
        if ai == 0:
            continue
        for j, bj in enumerate(b):
            k = i + j
            if k < N:
                out[k] += ai * bj
            else:
                out[k - N] -= ai * bj
    return [c % q for c in out]


def automorphism(a, galEl, q, logN):
    """X -> X^galEl on coefficient vectors, sign written as q - v (0 -> q like the reference)."""
    N = 1 << logN
    out = [0] * N
    for i in range(N):
        raw = i * galEl
        idx = raw & (N - 1)
        out[idx] = q - a[i] if (raw >> logN) & 1 else a[i]
    return out


# --------------------------------------------------------------------------- RNS helpers
def crt(residues, moduli):
    M = 1
    for m in moduli:
        M *= m
    x = 0
    for r, m in zip(residues, moduli):
        Mi = M // m
        x += r * Mi * pow(Mi, -1, m)
    return x % M, M


def moddown_exact(xq, xp, Q, P):
    """ModDownQPtoQ semantics: ((x mod Q) - [x]_P) / P mod q_i with [x]_P in [0,P): the value the
    reference computes whenever its float64 correction index v is the true floor."""
    lift, Pprod = crt(xp, P)
    return [((a - lift) * pow(Pprod, -1, q)) % q for a, q in zip(xq, Q)]


def modup_literal(src, Qs, Pt):
    """Literal restatement of modUpExact = reconstructRNS + multSum (basis_extension.go:337-357,
    537-646) for ONE coefficient: src residues under Qs -> lazy residues under Pt (in [0, ~3p))."""
    ns = len(Qs)
    Qprod = 1
    for q in Qs:
        Qprod *= q
    y = []
    vi = 0.0
    for i, q in enumerate(Qs):
        t = mform(pow(Qprod // q, -1, q), q)            # qoverqiinvqi[i]
        yi = mred(src[i], t, q, qinv64(q))
        y.append(yi)
        vi += float(yi) / float(q)
    v = int(vi)
    out = []
    for p in Pt:
        acc = 0
        for i, q in enumerate(Qs):
            acc += y[i] * mform((Qprod // q) % p, p)      # qoverqimodp[j][i]
        acc &= (1 << 128) - 1
        rhi, rlo = acc >> 64, acc & (R - 1)
        hhi = (((rlo * qinv64(p)) & (R - 1)) * p) >> 64
        vt = (v * (p - Qprod % p)) % p                    # vtimesqmodp[j][v]
        out.append((rhi - hhi + p + vt) & (R - 1))
    return out, v


def div_round_last(x, Q):
    """DivRoundByLastModulus: floor((x + h)/qL) with h = (qL-1)>>1, per RNS limb (App. A.6)."""
    L = len(Q) - 1
    qL = Q[L]
    h = (qL - 1) >> 1
    t = (x[L] + h) % qL
    return [((x[i] - (t - h)) * pow(qL, -1, Q[i])) % Q[i] for i in range(L)]


# --------------------------------------------------------------------------- multi-key model
class Model:
    """Mathematical model of mkrlwe.KeySwitcher for small N.  Polynomials are lists
    [limb][coeff] of Python ints; switching keys arrive in the reference's storage form
    (NTT domain, Montgomery) and are interpreted by definition."""

    def __init__(self, logN, Q, P, gamma=2, psiQ=None, psiP=None):
        self.logN, self.N = logN, 1 << logN
        self.Q, self.P = list(Q), list(P)
        self.QP = self.Q + self.P
        self.gamma = gamma
        self.alpha = len(P) // gamma
        self.psi = list(psiQ) if psiQ else [find_psi(q, self.N) for q in self.Q]
        self.psi += list(psiP) if psiP else [find_psi(p, self.N) for p in self.P]

    def beta(self, level):
        return -(-(level + 1) // self.alpha)

    def limb_index(self, level):
        """indices into QP (and into a [nQ+nP] switching-key poly) active at `level`."""
        return list(range(level + 1)) + [len(self.Q) + j for j in range(len(self.P))]

    def key_coeff(self, swk_poly, j):
        """storage form (NTT, Montgomery) limb j of a PolyQP -> plain coefficient vector."""
        q = self.QP[j]
        return intt_def([invmform(int(v), q) for v in swk_poly[j]], q, self.psi[j], self.logN)

    def digit_values(self, a, level, i):
        """integer value of digit i of polynomial a (coefficient domain), per coefficient."""
        lo = i * self.alpha
        hi = min(lo + self.alpha, level + 1)
        mods = self.Q[lo:hi]
        if hi - lo == 1:
            return [int(v) for v in a[lo]]
        return [crt([int(a[l][k]) for l in range(lo, hi)], mods)[0] for k in range(self.N)]

    def decompose(self, a, level):
        """h(a): [beta][limb in QP] NTT-domain canonical values (None for inactive limbs)."""
        out = []
        for i in range(self.beta(level)):
            vals = self.digit_values(a, level, i)
            row = [None] * len(self.QP)
            for j in self.limb_index(level):
                q = self.QP[j]
                row[j] = ntt_def([v % q for v in vals], q, self.psi[j], self.logN)
            out.append(row)
        return out

    def external_product(self, a, bg, level):
        """ModDown_P( sum_i g_i * digit_i(a) ), bg in storage form; `a` coefficient domain."""
        acc = {j: [0] * self.N for j in self.limb_index(level)}
        for i in range(self.beta(level)):
            vals = self.digit_values(a, level, i)
            for j in self.limb_index(level):
                q = self.QP[j]
                prod = negacyclic_mul([v % q for v in vals], self.key_coeff(bg[i], j), q)
                acc[j] = [(x + y) % q for x, y in zip(acc[j], prod)]
        return self._moddown(acc, level)

    def _moddown(self, acc, level):
        nq = len(self.Q)
        out = [[0] * self.N for _ in range(level + 1)]
        for k in range(self.N):
            xq = [acc[j][k] for j in range(level + 1)]
            xp = [acc[nq + j][k] for j in range(len(self.P))]
            r = moddown_exact(xq, xp, self.Q[: level + 1], self.P)
            for j in range(level + 1):
                out[j][k] = r[j]
        return out

    def _acc_vector(self, terms, level):
        """x = sum_t key_t (.) h(a_t) as plain coefficient-domain PolyQP per digit."""
        beta = self.beta(level)
        x = [{j: [0] * self.N for j in self.limb_index(level)} for _ in range(beta)]
        for a, key in terms:
            for i in range(beta):
                vals = self.digit_values(a, level, i)
                for j in self.limb_index(level):
                    q = self.QP[j]
                    prod = negacyclic_mul([v % q for v in vals], self.key_coeff(key[i], j), q)
                    x[i][j] = [(u + w) % q for u, w in zip(x[i][j], prod)]
        return x

    def _ext_plain(self, a, xvec, level):
        acc = {j: [0] * self.N for j in self.limb_index(level)}
        for i in range(self.beta(level)):
            vals = self.digit_values(a, level, i)
            for j in self.limb_index(level):
                q = self.QP[j]
                prod = negacyclic_mul([v % q for v in vals], xvec[i][j], q)
                acc[j] = [(u + w) % q for u, w in zip(acc[j], prod)]
        return self._moddown(acc, level)

    def mul_and_relin(self, level, ids0, op0, ids1, op1, rlk, crs_u):
        """SURVEY.md App. C.  op: [1+n][limbs][N]; rlk {id: (b, d, v)} storage form."""
        Q = self.Q[: level + 1]
        ids_out = sorted(set(ids0) | set(ids1))
        out = {o: [[0] * self.N for _ in Q] for o in ["0"] + ids_out}
        c0 = lambda op, s: [[int(v) for v in op[s][l]] for l in range(level + 1)]
        x = self._acc_vector([(c0(op0, 1 + a), rlk[i][1]) for a, i in enumerate(ids0)], level)
        y = self._acc_vector([(c0(op1, 1 + a), rlk[i][0]) for a, i in enumerate(ids1)], level)
        a0, b0 = c0(op0, 0), c0(op1, 0)
        for l, q in enumerate(Q):
            out["0"][l] = negacyclic_mul(a0[l], b0[l], q)
        for a, i in enumerate(ids0):
            ai = c0(op0, 1 + a)
            for l, q in enumerate(Q):
                out[i][l] = negacyclic_mul(b0[l], ai[l], q)
        for a, i in enumerate(ids1):
            bi = c0(op1, 1 + a)
            for l, q in enumerate(Q):
                pr = negacyclic_mul(a0[l], bi[l], q)
                out[i][l] = [(u + w) % q for u, w in zip(out[i][l], pr)] if i in ids0 else pr
        for a, j in enumerate(ids1):
            e = self._ext_plain(c0(op1, 1 + a), x, level)
            for l, q in enumerate(Q):
                out[j][l] = [(u + w) % q for u, w in zip(out[j][l], e[l])]
        for a, i in enumerate(ids0):
            t = self._ext_plain(c0(op0, 1 + a), y, level)
            e0 = self.external_product(t, rlk[i][2], level)
            e1 = self.external_product(t, crs_u, level)
            for l, q in enumerate(Q):
                out["0"][l] = [(u + w) % q for u, w in zip(out["0"][l], e0[l])]
                out[i][l] = [(u + w) % q for u, w in zip(out[i][l], e1[l])]
        return ids_out, [out["0"]] + [out[i] for i in ids_out]


# --------------------------------------------------------------------------- multi-key BFV model
class BfvModel(Model):
    """Mathematical model of mkbfv's multiplication path (mkbfv/basis_extension.go, keyswitch_hoisted.go,
    evaluator.go:118-140) for small N, alpha = 1.  Base conversions use the literal modUpExact of modup_literal
    (float64 correction index included) where the reference's result depends on it; products are schoolbook
    negacyclic products; nothing is shared with the C oracle."""

    def __init__(self, logN, Q, QMul, P, T):
        super().__init__(logN, Q, P, 2)
        assert self.alpha == 1 and len(Q) == len(QMul)
        self.QMul, self.T = list(QMul), T
        self.R = self.Q + self.QMul
        self.psiR = self.psi[: len(Q)] + [find_psi(q, self.N) for q in self.QMul]
        self.Qprod = 1
        for q in self.Q:
            self.Qprod *= q
        self.QMulprod = 1
        for q in self.QMul:
            self.QMulprod *= q

    # ---- FastBasisExtender (coefficient domain, one polynomial = [limb][coeff])
    def _cols(self, poly, n):
        return [[int(poly[l][k]) for l in range(n)] for k in range(self.N)]

    def modup_q_to_r(self, polyq):
        nq = len(self.Q)
        out = [[int(v) for v in polyq[l]] for l in range(nq)] + [[0] * self.N for _ in range(nq)]
        for k, col in enumerate(self._cols(polyq, nq)):
            lift, _ = modup_literal(col, self.Q, self.QMul)
            for j in range(nq):
                out[nq + j][k] = lift[j]
        return out

    def rescale(self, polyq):
        nq = len(self.Q)
        out = [[0] * self.N for _ in range(2 * nq)]
        for k, col in enumerate(self._cols(polyq, nq)):
            pq = [(col[i] * (self.QMulprod % q)) % q for i, q in enumerate(self.Q)]
            lift, _ = modup_literal(pq, self.Q, self.QMul)
            qm = [((0 - lift[j]) * pow(self.Qprod, -1, p)) % p for j, p in enumerate(self.QMul)]      # ModDownQPtoP
            back, _ = modup_literal(qm, self.QMul, self.Q)                                              # ModUpPtoQ (lazy)
            for j in range(nq):
                out[j][k], out[nq + j][k] = back[j], qm[j]
        return out

    def quantize_coeff(self, polyr):
        """Quantize of a COEFFICIENT-domain polynomial over R (the reference passes the NTT form and inverts it)."""
        nq = len(self.Q)
        out = [[0] * self.N for _ in range(nq)]
        for k in range(self.N):
            tq = [(int(polyr[l][k]) * self.T) % self.Q[l] for l in range(nq)]
            tm = [(int(polyr[nq + l][k]) * self.T) % self.QMul[l] for l in range(nq)]
            lift, _ = modup_literal(tm, self.QMul, self.Q)
            for i, q in enumerate(self.Q):
                out[i][k] = ((tq[i] - lift[i]) * pow(self.QMulprod, -1, q)) % q                         # ModDownQPtoQ
        return out

    # ---- gadget pieces
    def _acc_digits(self, terms):
        """sum_t key_t (.) NTT(digits_t): terms = [(digit value lists [nd][N], key [nd][m][N])] -> [nd]{j: coeffs}"""
        nd = len(terms[0][0])
        idx = self.limb_index(len(self.Q) - 1)
        x = [{j: [0] * self.N for j in idx} for _ in range(nd)]
        for digs, key in terms:
            for i in range(nd):
                for j in idx:
                    q = self.QP[j]
                    prod = negacyclic_mul([v % q for v in digs[i]], self.key_coeff(key[i], j), q)
                    x[i][j] = [(u + w) % q for u, w in zip(x[i][j], prod)]
        return x

    def _ext_digits(self, pairs):
        """ModDown_P( sum over (digits, xvec) pairs, sum_i xvec[i] * digit_i )"""
        level = len(self.Q) - 1
        acc = {j: [0] * self.N for j in self.limb_index(level)}
        for digs, xvec in pairs:
            for i in range(len(digs)):
                for j in self.limb_index(level):
                    q = self.QP[j]
                    prod = negacyclic_mul([v % q for v in digs[i]], xvec[i][j], q)
                    acc[j] = [(u + w) % q for u, w in zip(acc[j], prod)]
        return self._moddown(acc, level)

    def mul_relin_new(self, ids0, op0, ids1, op1, rlk, crs_u):
        """Evaluator.MulRelinNew; rlk {id: (b1, b2, d1, d2, v)} in storage form -> (ids_out, [1+n][nQ][N])"""
        nq, level = len(self.Q), len(self.Q) - 1
        c0R = [self.modup_q_to_r(op0[s]) for s in range(1 + len(ids0))]
        c1R = [self.rescale(op1[s]) for s in range(1 + len(ids1))]
        dq = lambda pr: ([pr[l] for l in range(nq)], [pr[nq + l] for l in range(nq)])     # (Q digits, QMul digits)
        x1 = self._acc_digits([(dq(c0R[1 + a])[0], rlk[i][2]) for a, i in enumerate(ids0)])
        x2 = self._acc_digits([(dq(c0R[1 + a])[1], rlk[i][3]) for a, i in enumerate(ids0)])
        y1 = self._acc_digits([(dq(c1R[1 + a])[0], rlk[i][0]) for a, i in enumerate(ids1)])
        y2 = self._acc_digits([(dq(c1R[1 + a])[1], rlk[i][1]) for a, i in enumerate(ids1)])
        ids_out = sorted(set(ids0) | set(ids1))
        mulR = lambda a, b: [negacyclic_mul([v % m for v in a[l]], [v % m for v in b[l]], m) for l, m in enumerate(self.R)]
        addR = lambda a, b: [[(u + w) % m for u, w in zip(a[l], b[l])] for l, m in enumerate(self.R)]
        out = {"0": self.quantize_coeff(mulR(c0R[0], c1R[0]))}
        for i in ids_out:
            acc = None
            if i in ids0:
                acc = mulR(c1R[0], c0R[1 + ids0.index(i)])
            if i in ids1:
                t = mulR(c0R[0], c1R[1 + ids1.index(i)])
                acc = t if acc is None else addR(acc, t)
            out[i] = self.quantize_coeff(acc)
        Q = self.Q
        for a, j in enumerate(ids1):
            d1, d2 = dq(c1R[1 + a])
            e = self._ext_digits([(d1, x1), (d2, x2)])
            for l, q in enumerate(Q):
                out[j][l] = [(u + w) % q for u, w in zip(out[j][l], e[l])]
        for a, i in enumerate(ids0):
            d1, d2 = dq(c0R[1 + a])
            t = self._ext_digits([(d1, y1), (d2, y2)])
            e0 = self.external_product(t, rlk[i][4], level)
            e1 = self.external_product(t, crs_u, level)
            for l, q in enumerate(Q):
                out["0"][l] = [(u + w) % q for u, w in zip(out["0"][l], e0[l])]
                out[i][l] = [(u + w) % q for u, w in zip(out[i][l], e1[l])]
        return ids_out, [out["0"]] + [out[i] for i in ids_out]
