"""TEST INFRASTRUCTURE -- CPU restatement of the refit buffer's shuffled split.

The reference pools the rows of one outer iteration, shuffles them with `torch.randperm`, cuts at `train_pct` and caps both
parts (`nfmc/algorithms/sampling/tuning.py:44-65`).  The build's device path (`csrc/fit_support.hip: nfmc_rows_sample_f32`)
gathers rows pi(0), pi(1), ... of a keyed pseudo-random permutation pi of the pooled rows instead of sorting N random keys:
positions [0, n_train) and [cut, cut + n_val) of a uniform shuffle are, in distribution, any n_train + n_val distinct
positions of it.  The stream is the build's own (as the Philox noise is): parity with the reference is distributional, the
permutation itself is pinned HERE, bit for bit, against the kernel and the library's host evaluation of it.

pi: balanced Feistel network on 2 * half bits (half = ceil(ceil(log2 N) / 2), at least 1), 6 rounds, round function
mix32(R + key_r) masked to `half` bits, cycle-walking until the value is < N; keys = the first 6 outputs of splitmix64(seed).
"""
import numpy as np

ROUNDS = 6
_M32 = 0xFFFFFFFF
_M64 = 0xFFFFFFFFFFFFFFFF


def _mix32(x):
    x &= _M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & _M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & _M32
    x ^= x >> 16
    return x


def keys(n, seed):
    bits = 2
    while bits < 62 and (1 << bits) < n:
        bits += 1
    half = (bits + 1) // 2
    s, ks = seed & _M64, []
    for _ in range(ROUNDS):
        s = (s + 0x9E3779B97F4A7C15) & _M64
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        z ^= z >> 31
        ks.append(z & _M32)
    return half, ks


def index(i, n, seed):
    """pi(i) for 0 <= i < n."""
    half, ks = keys(n, seed)
    mask = (1 << half) - 1
    while True:
        L, R = i >> half, i & mask
        for k in ks:
            L, R = R, L ^ (_mix32(R + k) & mask)
        i = (L << half) | R
        if i < n:
            return i


def permutation_prefix(n, seed, m, first=0):
    return np.array([index(first + i, n, seed) for i in range(m)], dtype=np.int64)


def train_val_split(x, train_pct, max_train_size, max_val_size, seed):
    """(n_iterations, n_chains, *event) -> (x_train, x_val) the way `nfmc_amd.tuning.train_val_split` cuts it on the device."""
    rows = x.reshape((-1,) + tuple(x.shape[2:]))
    total = rows.shape[0]
    cut = int(train_pct * total)
    n_train, n_val = min(cut, int(max_train_size)), min(total - cut, int(max_val_size))
    idx = permutation_prefix(total, seed, n_train + n_val)
    return rows[idx[:n_train]], rows[idx[n_train:]]
