"""ORACLE (test infrastructure only — never imported by the product path).

numpy restatement of csrc/sampler.hip, i.e. of the negative sampling the reference does in
``beta_rec/data/base_data.py:218-253`` (instance_bpr_loader), ``:182-216`` (instance_bce_loader) and
``:254-288`` (instance_mul_neg_loader): per training row, ``random.sample(list(set(item_id_pool) -
positive_items(user)), k)`` — k distinct items, uniform over the items the user never touched.

Two layers:
* ``missing_item(positives, r)`` — the r-th item (ascending) NOT in a user's positive set: the map
  from a uniform rank to an item, restated the obvious way (set difference, sort, index);
* ``sample_negatives`` — the counter-based generator of the kernel (splitmix64 row key + 6-round
  Feistel bijection with cycle walking) bit for bit, so device output can be checked exactly.
Python's Mersenne-Twister stream and set iteration order are not reproducible on a GPU; against the
reference itself the tests check support / distinctness / uniformity on tests/golden/sampler_*.npz.
"""
import numpy as np

U64 = np.uint64
U32 = np.uint32


def splitmix64(x):
    with np.errstate(over="ignore"):
        x = np.asarray(x, dtype=U64) + U64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> U64(30))) * U64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> U64(27))) * U64(0x94D049BB133111EB)
        return x ^ (x >> U64(31))


def feistel_half_bits(n):
    bits = 2
    while bits < 62 and (1 << bits) < n:
        bits += 1
    return (bits + 1) // 2


def _feistel_round(x, key):
    with np.errstate(over="ignore"):
        x = (x ^ key) * U32(0x9E3779B1)
        x = x ^ (x >> U32(15))
        x = x * U32(0x85EBCA77)
        return x ^ (x >> U32(13))


def feistel_permute(i, n, seed):
    """P_seed(i) over [0, n) for arrays ``i``, ``n``, ``seed`` of equal length (n may differ per element)."""
    i = np.asarray(i, dtype=U64).copy()
    n = np.broadcast_to(np.asarray(n, dtype=U64), i.shape)
    seed = np.broadcast_to(np.asarray(seed, dtype=U64), i.shape)
    half = np.array([feistel_half_bits(int(v)) for v in n], dtype=U64)
    mask = (U64(1) << half) - U64(1)
    x = i
    todo = np.ones(i.shape, dtype=bool)
    while todo.any():
        xs, hs, ms, ss = x[todo], half[todo], mask[todo], seed[todo]
        left = (xs >> hs).astype(U32)
        right = (xs & ms).astype(U32)
        for rnd in range(6):
            with np.errstate(over="ignore"):
                key = ((ss >> U64(8 * (rnd & 3))).astype(U32) + U32(0x632BE5AB) * U32(rnd + 1)
                       + (ss >> U64(32)).astype(U32))
            f = _feistel_round(right, key) & ms.astype(U32)
            left, right = right, left ^ f
        xs = (left.astype(U64) << hs) | right.astype(U64)
        x[todo] = xs
        again = xs >= n[todo]
        idx = np.flatnonzero(todo)
        todo[idx[~again]] = False
    return x


def random_permutation(n, seed):
    """hiprec_random_permutation (csrc/util.hip): out[i] = P_seed(i)."""
    return feistel_permute(np.arange(n, dtype=U64), np.full(n, n, dtype=U64), np.full(n, seed, dtype=U64)).astype(np.int64)


def missing_item(positives, r, n_items):
    """r-th (0-based, ascending) item of range(n_items) that is not in ``positives``."""
    return sorted(set(range(n_items)) - set(int(p) for p in positives))[r]


def positive_csr(train_users, train_items, n_users):
    """(user_ptr, pos_sorted): ascending unique positives per user, base_data.py:227-231."""
    sets = [set() for _ in range(n_users)]
    for u, i in zip(train_users, train_items):
        sets[int(u)].add(int(i))
    ptr = np.zeros(n_users + 1, dtype=np.int64)
    ptr[1:] = np.cumsum([len(s) for s in sets])
    cols = np.array([i for s in sets for i in sorted(s)], dtype=np.int64)
    return ptr, cols


def sample_negatives(train_users, train_items, n_users, n_items, k, seed):
    """[n_rows, k] negatives exactly as the kernel draws them."""
    ptr, cols = positive_csr(train_users, train_items, n_users)
    users = np.asarray(train_users, dtype=np.int64)
    n_rows = users.size
    rows = np.repeat(np.arange(n_rows, dtype=U64), k)
    js = np.tile(np.arange(k, dtype=U64), n_rows)
    deg = (ptr[1:] - ptr[:-1])[users]
    m = np.repeat(n_items - deg, k).astype(U64)
    if (m < U64(k)).any():
        raise ValueError("Sample larger than population or is negative")
    row_seed = splitmix64(U64(seed) ^ splitmix64(rows))
    ranks = feistel_permute(js, m, row_seed).astype(np.int64)
    out = np.empty(n_rows * k, dtype=np.int64)
    for idx, (row, r) in enumerate(zip(rows.astype(np.int64), ranks)):
        u = users[row]
        out[idx] = missing_item(cols[ptr[u]:ptr[u + 1]], int(r), n_items)
    return out.reshape(n_rows, k)
