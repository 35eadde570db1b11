"""Pure-Python (math module, plain loops) restatement of the reference's small
routines, written from the reference text independently of oracle/*.c, for SMALL
cases only.  It cross-checks the C oracle and provides brute-force answers.

Cited lines are into /root/reference."""
import itertools
import math

INF = 1e15       # shared/gen_func.hpp:15
EPSILON = 1e-5   # shared/gen_func.hpp:16


def _log(x):
    if x == 0:
        return -math.inf
    if x < 0 or x != x:
        return math.nan
    return math.log(x)


def _exp(x):
    try:
        return math.exp(x)
    except OverflowError:
        return math.inf


def logsum(a):  # gen_func.cpp:135-151
    M = a[0]
    for v in a[1:]:
        M = v if v >= M else M
    if M == -math.inf:
        return -math.inf
    s = 0.0
    for v in a:
        s += _exp(v - M)
    return _log(s) + M


def calc_trans(k, l, q_l, alpha, d):  # HMM.cpp:130-139
    c = _exp(-alpha * d)
    t = (1 - c) * q_l
    if k == l:
        t += c
    return _log(t)


def calc_hwe(maf, F, log_scale=True):  # gen_func.cpp:938-957
    g = [(1 - maf) * (1 - maf) + (1 - maf) * maf * F,
         2 * (1 - maf) * maf - 2 * (1 - maf) * maf * F,
         maf * maf + (1 - maf) * maf * F]
    if log_scale:
        g = [_log(v) for v in g]
        g = [-INF if v == -math.inf else v for v in g]
    if F == 1:
        g[1] = -INF if log_scale else 1 / INF
    return g


def post_prob(lkl, prior):  # gen_func.cpp:920-932
    pp = [lkl[g] + (prior[g] if prior is not None else 0.0) for g in range(3)]
    norm = logsum(pp)
    return [v - norm for v in pp]


def calc_emission(gl, maf, k):  # HMM.cpp:144-154
    h = calc_hwe(maf, float(k))
    return logsum([gl[g] + h[g] for g in range(3)])


def est_maf(gl_site, indF):  # gen_func.cpp:974-1009
    iters = 0
    passes = 0
    num = den = 0.0
    freq = 0.01
    while True:
        prev = freq
        passes += 1
        for i in range(len(indF)):
            F = indF[i]
            pp = post_prob(gl_site[i], calc_hwe(freq, F))
            pp = [_exp(v) for v in pp]
            num += pp[1] + pp[2] * (2 - F)
            den += 2 * pp[1] + (pp[0] + pp[2]) * (2 - F)
        freq = num / den
        cont = abs(prev - freq) > EPSILON
        if cont:
            cont = iters < 100
            iters += 1
        if not cont:
            break
    return freq, passes


def forward(q, alpha, e, pos):  # HMM.cpp:6-28
    S = len(pos)
    Fw = [[_log(q[0]), _log(q[1])]]
    for s in range(1, S + 1):
        row = []
        for l in range(2):
            tmp = [Fw[s - 1][k] + calc_trans(k, l, q[l], alpha, pos[s - 1]) for k in range(2)]
            row.append(logsum(tmp) + e[s - 1][l])
        Fw.append(row)
    return logsum(Fw[S]), Fw


def backward(q, alpha, e, pos):  # HMM.cpp:33-60
    S = len(pos)
    Bw = [[0.0, 0.0] for _ in range(S + 1)]
    for s in range(S, 0, -1):
        for k in range(2):
            tmp = [calc_trans(k, l, q[l], alpha, pos[s - 1]) + e[s - 1][l] + Bw[s][l]
                   for l in range(2)]
            Bw[s - 1][k] = logsum(tmp)
    Bw0 = [Bw[0][k] + _log(q[k]) for k in range(2)]
    full = [Bw0] + Bw[1:]
    return logsum(Bw0), full


def check_interv(v):  # gen_func.cpp:55-70
    if v < EPSILON:
        return 0.0
    if v > 1 - EPSILON:
        return 1.0
    return v


def posteriors(q, alpha, e, pos):  # EM.cpp:178-185
    lkl, Fw = forward(q, alpha, e, pos)
    _, Bw = backward(q, alpha, e, pos)
    S = len(pos)
    return [[check_interv(_exp(Bw[s][k] + Fw[s][k] - lkl)) for k in range(2)]
            for s in range(1, S + 1)]


def viterbi(q, alpha, e, pos):  # HMM.cpp:98-125 (with its in-place update of Vi_prob)
    S = len(pos)
    V = [_log(q[0]), _log(q[1])]
    back = [[0, 0] for _ in range(S + 1)]
    for s in range(1, S + 1):
        for l in range(2):
            vmax, kmax = -INF, 0
            for k in range(2):
                pval = V[k] + calc_trans(k, l, q[l], alpha, pos[s - 1])
                if vmax < pval:
                    vmax, kmax = pval, k
            back[s][l] = kmax
            V[l] = vmax + e[s - 1][l]
    path = [0] * (S + 1)
    best, mx = 0, -math.inf
    for c in range(2):
        if V[c] > mx:
            best, mx = c, V[c]
    path[S] = best
    for s in range(S, 0, -1):
        path[s - 1] = back[s][path[s]]
    return path


def brute_force_loglik(q, alpha, e, pos):
    """log sum over all 2^(S+1) state paths z_0..z_S of q[z0] prod T_s e_s: the
    likelihood the forward recursion must equal (closed form, no recursion)."""
    S = len(pos)
    tot = 0.0
    for z in itertools.product((0, 1), repeat=S + 1):
        p = q[z[0]]
        for s in range(1, S + 1):
            p *= math.exp(calc_trans(z[s - 1], z[s], q[z[s]], alpha, pos[s - 1]) + e[s - 1][z[s]])
        tot += p
    return math.log(tot)


def brute_force_posterior(q, alpha, e, pos):
    S = len(pos)
    tot = 0.0
    m = [0.0] * S
    for z in itertools.product((0, 1), repeat=S + 1):
        p = q[z[0]]
        for s in range(1, S + 1):
            p *= math.exp(calc_trans(z[s - 1], z[s], q[z[s]], alpha, pos[s - 1]) + e[s - 1][z[s]])
        tot += p
        for s in range(S):
            if z[s + 1] == 1:
                m[s] += p
    return [v / tot for v in m]


class Taus:
    """GSL's gsl_rng_taus (L'Ecuyer 1996 three-component Tausworthe generator), the only GSL
    generator the reference uses (parse_args.cpp:232-233), restated from the published
    recurrence; known answer (GSL's own test): seed 1, 10 000th output 2733957125."""
    M = 0xFFFFFFFF

    def __init__(self, seed):
        s = seed & self.M
        if s == 0:
            s = 1
        self.s1 = (69069 * s) & self.M
        self.s2 = (69069 * self.s1) & self.M
        self.s3 = (69069 * self.s2) & self.M
        for _ in range(6):
            self.next()

    @staticmethod
    def _step(s, a, b, c, d):
        M = 0xFFFFFFFF
        return ((((s & c) << d) & M) ^ ((((s << a) & M) ^ s) >> b)) & M

    def next(self):
        self.s1 = self._step(self.s1, 13, 19, 4294967294, 12)
        self.s2 = self._step(self.s2, 2, 25, 4294967288, 4)
        self.s3 = self._step(self.s3, 3, 11, 4294967280, 17)
        return self.s1 ^ self.s2 ^ self.s3

    def uniform(self):
        return self.next() / 4294967296.0


def random_initial_values(seed, n_ind, n_sites, indF_random=True, freq_random=True):
    """parse_args.cpp:248-254 and :305-310: one generator; indF_i then alpha_i per individual
    in [1e-6, 1 - 1e-6], then one frequency per site in [0.01, 0.49]."""
    rng = Taus(seed)
    f_min, f_max = 0.000001, 1 - 0.000001
    indF, alpha = [], []
    if indF_random:
        for _ in range(n_ind):
            indF.append(f_min + rng.uniform() * (f_max - f_min))
            alpha.append(f_min + rng.uniform() * (f_max - f_min))
    q_min = 0.01
    q_max = 0.5 - q_min
    freq = [q_min + rng.uniform() * (q_max - q_min) for _ in range(n_sites)] if freq_random else []
    return indF, alpha, freq
