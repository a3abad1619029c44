"""MovieLens-1M files -> graph inputs, the preprocessing of the reference's
training script (``train_movielens.py:29-151``) restated with numpy and the
standard library (the script itself relies on pandas-1.x positional arguments
and runs at import).

Files are ``::``-separated; ``movies.dat`` is ISO-8859-1.  ``ratings.dat`` is
not distributed with the reference, the caller supplies its path.

What the script does, step by step, and what is kept:

* movies: the ``|``-separated genre field becomes a multi-hot vector over the
  genre vocabulary (``:29-50``).  The script numbers the vocabulary by iterating
  a ``set`` of strings, an order that changes from process to process; here the
  vocabulary is sorted.
* users: sex, occupation (as a string) and a 5-bin age group are one-hot encoded
  in that order (``:58-66``); the columns of each group follow the sorted
  category strings (pandas ``get_dummies``), so occupations run "0", "1", "10",
  "11", ..., and the age group keeps all five bins of
  ``(0,20], (20,30], (30,40], (40,50], (50,60]`` whether used or not.
* ratings: <= 2 becomes label 0, >= 5 label 1, the rest is dropped; exact
  duplicate rows are dropped (``:77-87``).  Users and movies without a positive
  rating are removed everywhere (``:89-94``).
* ids: users and movies are renumbered by the rank of the STRINGS ``u_<id>`` /
  ``i_<id>`` (``:98-111``), i.e. lexicographically ("u_10" < "u_2"); movies are
  then shifted behind the users (``:114-116``).
* feature rows stay in FILE order (``:120-123``) while node ids follow the
  string rank, so in the reference row ``k`` of a feature table is in general not
  the row of node ``k``.  ``align_features=False`` keeps that; the default
  ``True`` puts the row of node ``k`` at position ``k``.
* graph edges are the distinct positive (user, movie) pairs, then their flipped
  copies (``:130-132``); links for training are all kept ratings with their
  labels (``:141-151``).
"""
import numpy as np

__all__ = ['read_dat', 'user_features', 'movie_features', 'binary_label', 'load_movielens', 'train_test_split',
           'synthetic_ratings']

AGE_BINS = (0, 20, 30, 40, 50, 60)


def read_dat(path, encoding='utf-8'):
    """Rows of a ``::``-separated file as lists of strings."""
    rows = []
    with open(path, 'r', encoding=encoding) as fh:
        for line in fh:
            line = line.rstrip('\r\n')
            if line:
                rows.append(line.split('::'))
    return rows


def _one_hot(values, categories):
    index = {c: i for i, c in enumerate(categories)}
    out = np.zeros((len(values), len(categories)), dtype=np.float32)
    for r, v in enumerate(values):
        if v in index:
            out[r, index[v]] = 1.0
    return out


def _age_bin(age):
    """Index of the right-closed bin holding ``age`` (``pandas.cut`` semantics), or None."""
    for i in range(len(AGE_BINS) - 1):
        if AGE_BINS[i] < age <= AGE_BINS[i + 1]:
            return i
    return None


def user_features(rows):
    """(user ids int64[n], features float32[n, 2 + occupations + 5], column names)
    from users.dat rows ``user_id::sex::age::occupation::post_code``."""
    ids = np.asarray([int(r[0]) for r in rows], dtype=np.int64)
    sex = [r[1] for r in rows]
    occupation = [str(int(r[3])) for r in rows]
    age = [_age_bin(int(r[2])) for r in rows]
    sex_cats = sorted(set(sex))
    occ_cats = sorted(set(occupation))                    # string order, as get_dummies sorts them
    age_cats = list(range(len(AGE_BINS) - 1))
    feats = np.concatenate([_one_hot(sex, sex_cats), _one_hot(occupation, occ_cats), _one_hot(age, age_cats)], axis=1)
    names = ['sex_' + c for c in sex_cats] + ['occupation_' + c for c in occ_cats] + \
            ['age_bin_%d' % c for c in age_cats]
    return ids, feats, names


def movie_features(rows):
    """(movie ids int64[n], multi-hot genre features float32[n, genres], genre names)
    from movies.dat rows ``movie_id::title::genre|genre|...``."""
    ids = np.asarray([int(r[0]) for r in rows], dtype=np.int64)
    tags = [r[2].split('|') for r in rows]
    vocab = sorted(set(t for doc in tags for t in doc))
    index = {t: i for i, t in enumerate(vocab)}
    feats = np.zeros((len(rows), len(vocab)), dtype=np.float32)
    for r, doc in enumerate(tags):
        for t in doc:
            feats[r, index[t]] = 1.0
    return ids, feats, vocab


def binary_label(rating):
    """<= 2 -> 0, >= 5 -> 1, otherwise -1 (dropped)."""
    rating = np.asarray(rating)
    return np.where(rating <= 2, 0, np.where(rating >= 5, 1, -1)).astype(np.int64)


def _string_rank(ids, prefix):
    """raw id -> rank of the string prefix_<id> among the distinct ids."""
    uniq = sorted({int(i) for i in ids}, key=lambda v: '%s_%d' % (prefix, v))
    return {v: k for k, v in enumerate(uniq)}


def load_movielens(users_path, movies_path, ratings, align_features=True, pad_to=None):
    """Graph inputs from the MovieLens files.

    ``ratings`` is the path of ratings.dat or an int array [n, 3] of
    (user id, movie id, rating).  Returns a dict with ``node_features``
    {'user','item'} float32, ``nodes`` {'user','item'} id lists, ``edges``
    int64 (2, 2P) both directions, ``links`` int64 [n, 2] (user node, item node)
    and ``labels`` int64 [n] of every kept rating, ``user_columns`` /
    ``item_columns`` names.  ``pad_to``: feature widths are rounded up to a
    multiple of it with zero columns (the fused kernels take multiples of 32).
    """
    u_ids, u_feat, u_cols = user_features(read_dat(users_path))
    m_ids, m_feat, m_cols = movie_features(read_dat(movies_path, encoding='iso-8859-1'))
    if isinstance(ratings, str):
        rat = np.asarray([[int(r[0]), int(r[1]), int(r[2])] for r in read_dat(ratings)], dtype=np.int64)
    else:
        rat = np.asarray(ratings, dtype=np.int64)[:, :3]
    rat = rat.reshape(-1, 3)

    y = binary_label(rat[:, 2])
    rat = np.stack([rat[:, 0], rat[:, 1], y], axis=1)[y >= 0]
    # exact duplicates dropped, first occurrence kept, file order preserved
    _, first = np.unique(rat, axis=0, return_index=True)
    rat = rat[np.sort(first)]

    pos = rat[rat[:, 2] == 1]
    valid_users, valid_items = set(pos[:, 0].tolist()), set(pos[:, 1].tolist())
    keep_u = np.asarray([int(i) in valid_users for i in u_ids], dtype=bool)
    keep_m = np.asarray([int(i) in valid_items for i in m_ids], dtype=bool)
    u_ids, u_feat = u_ids[keep_u], u_feat[keep_u]
    m_ids, m_feat = m_ids[keep_m], m_feat[keep_m]
    in_tables = np.asarray([int(a) in valid_users and int(b) in valid_items for a, b in rat[:, :2]], dtype=bool)
    rat = rat[in_tables.reshape(-1)] if len(rat) else rat

    u_rank, m_rank = _string_rank(u_ids, 'u'), _string_rank(m_ids, 'i')
    missing = [a for a in rat[:, 0].tolist() if a not in u_rank] + [b for b in rat[:, 1].tolist() if b not in m_rank]
    if missing:
        raise KeyError("ratings name ids absent from users.dat / movies.dat: %s" % sorted(set(missing))[:5])
    user_node = np.asarray([u_rank[int(a)] for a in rat[:, 0]], dtype=np.int64)
    n_users = (int(user_node.max()) + 1) if len(user_node) else 0
    item_node = np.asarray([m_rank[int(b)] for b in rat[:, 1]], dtype=np.int64) + n_users
    labels = rat[:, 2].copy()

    if align_features:
        order_u = np.argsort([u_rank[int(i)] for i in u_ids], kind='stable')
        order_m = np.argsort([m_rank[int(i)] for i in m_ids], kind='stable')
        u_feat, m_feat = u_feat[order_u], m_feat[order_m]
    if pad_to:
        def pad(a):
            w = -(-a.shape[1] // pad_to) * pad_to
            return np.concatenate([a, np.zeros((a.shape[0], w - a.shape[1]), dtype=a.dtype)], axis=1)
        u_feat, m_feat = pad(u_feat), pad(m_feat)

    links = np.stack([user_node, item_node], axis=1) if len(user_node) else np.zeros((0, 2), dtype=np.int64)
    pos_links = links[labels == 1]
    _, first = np.unique(pos_links, axis=0, return_index=True)
    pairs = pos_links[np.sort(first)].T
    edges = np.concatenate([pairs, pairs[::-1]], axis=1)
    nodes = {'user': sorted(u_rank.values()), 'item': sorted(v + n_users for v in m_rank.values())}
    return {'node_features': {'user': np.ascontiguousarray(u_feat), 'item': np.ascontiguousarray(m_feat)},
            'nodes': nodes, 'edges': edges, 'links': links, 'labels': labels,
            'user_columns': u_cols, 'item_columns': m_cols}


def train_test_split(n, train_size=0.7, test_size=0.3, seed=None):
    """Shuffled index split (the script uses scikit-learn's function of this name
    with these fractions, ``train_movielens.py:141-142``): (train idx, test idx)."""
    perm = np.random.default_rng(seed).permutation(n)
    n_test = int(np.ceil(test_size * n))
    n_train = int(np.floor(train_size * n))
    return perm[n_test:n_test + n_train], perm[:n_test]


def synthetic_ratings(users_path, movies_path, n_ratings=200_000, seed=0):
    """MovieLens-shaped stand-in for the missing ratings.dat: int64 [n, 3] of
    (user id, movie id, rating 1..5).  A rating leans on a hidden affinity between
    the user's row and the movie's genres, so there is something to learn."""
    u_ids, u_feat, _ = user_features(read_dat(users_path))
    m_ids, m_feat, _ = movie_features(read_dat(movies_path, encoding='iso-8859-1'))
    rng = np.random.default_rng(seed)
    mix = rng.standard_normal((u_feat.shape[1], m_feat.shape[1]))
    pop = rng.permutation(np.arange(1, len(m_ids) + 1, dtype=np.float64) ** -0.9)     # long-tailed popularity
    ui = rng.integers(0, len(u_ids), n_ratings)
    mi = rng.choice(len(m_ids), size=n_ratings, p=pop / pop.sum())
    score = np.einsum('nf,fg,ng->n', u_feat[ui], mix, m_feat[mi]) / np.sqrt(np.maximum(m_feat[mi].sum(1), 1.0))
    score = score + 0.5 * rng.standard_normal(n_ratings)
    cuts = np.quantile(score, [0.1, 0.25, 0.55, 0.8])
    stars = 1 + np.searchsorted(cuts, score)
    return np.stack([u_ids[ui], m_ids[mi], stars], axis=1).astype(np.int64)
