"""MovieLens preprocessing (host only) on hand-checked toy files, pandas as the
arbiter of its own get_dummies / cut conventions, and -- when the reference's
data directory is present -- the real users.dat / movies.dat.

ratings.dat is absent from the reference, so the end-to-end numbers of its
script are unpinned; these tests pin the preprocessing rules themselves.
"""
import os

import numpy as np
import pytest

from graphsage_amd import movielens

USERS = "1::F::1::10::48067\n2::M::56::16::70072\n10::M::25::0::55117\n3::F::45::2::02460\n"
MOVIES = ("1::Toy Story (1995)::Animation|Children's|Comedy\n2::Jumanji (1995)::Adventure|Children's\n"
          "10::Heat (1995)::Action\n3::Caf\xe9 (1995)::Comedy\n")
RATINGS = ("1::1::5::978300760\n1::2::1::978300761\n2::1::5::978300762\n2::10::3::978300763\n10::3::5::978300764\n"
           "10::1::2::978300765\n3::2::1::978300766\n1::1::5::978300760\n2::2::5::978300767\n2::3::4::978300768\n")


@pytest.fixture
def toy(tmp_path):
    (tmp_path / 'users.dat').write_text(USERS)
    (tmp_path / 'movies.dat').write_bytes(MOVIES.encode('iso-8859-1'))
    (tmp_path / 'ratings.dat').write_text(RATINGS)
    return tmp_path


def test_toy_files_give_the_hand_computed_graph_inputs(toy):
    d = movielens.load_movielens(str(toy / 'users.dat'), str(toy / 'movies.dat'), str(toy / 'ratings.dat'))
    # users 1, 10, 2 rank as the strings u_1 < u_10 < u_2; user 3 and movie 10 have no positive rating
    assert d['links'].tolist() == [[0, 3], [0, 4], [2, 3], [1, 5], [1, 3], [2, 4]]
    assert d['labels'].tolist() == [1, 0, 1, 1, 0, 1]
    assert d['edges'].tolist() == [[0, 2, 1, 2, 3, 3, 5, 4], [3, 3, 5, 4, 0, 2, 1, 2]]
    assert d['nodes'] == {'user': [0, 1, 2], 'item': [3, 4, 5]}
    assert d['user_columns'] == ['sex_F', 'sex_M', 'occupation_0', 'occupation_10', 'occupation_16', 'occupation_2',
                                 'age_bin_0', 'age_bin_1', 'age_bin_2', 'age_bin_3', 'age_bin_4']
    assert d['item_columns'] == ['Action', 'Adventure', 'Animation', "Children's", 'Comedy']
    u1 = [1, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0]
    u2 = [0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1]
    u10 = [0, 1, 1, 0, 0, 0, 0, 1, 0, 0, 0]
    assert d['node_features']['user'].tolist() == [u1, u10, u2]                 # row k = node k
    assert d['node_features']['item'].tolist() == [[0, 0, 1, 1, 1], [0, 1, 0, 1, 0], [0, 0, 0, 0, 1]]
    assert d['node_features']['user'].dtype == np.float32
    ref = movielens.load_movielens(str(toy / 'users.dat'), str(toy / 'movies.dat'), str(toy / 'ratings.dat'),
                                   align_features=False)
    assert ref['node_features']['user'].tolist() == [u1, u2, u10]               # file order, as the script leaves it
    padded = movielens.load_movielens(str(toy / 'users.dat'), str(toy / 'movies.dat'), str(toy / 'ratings.dat'),
                                      pad_to=32)
    assert padded['node_features']['user'].shape == (3, 32) and padded['node_features']['item'].shape == (3, 32)
    assert padded['node_features']['user'][:, :11].tolist() == [u1, u10, u2]
    assert not padded['node_features']['user'][:, 11:].any()
    # array input instead of a path
    arr = np.asarray([[int(x) for x in l.split('::')[:3]] for l in RATINGS.strip().split('\n')])
    d2 = movielens.load_movielens(str(toy / 'users.dat'), str(toy / 'movies.dat'), arr)
    assert d2['links'].tolist() == d['links'].tolist()
    with pytest.raises(KeyError):
        movielens.load_movielens(str(toy / 'users.dat'), str(toy / 'movies.dat'), np.asarray([[77, 1, 5]]))


def test_label_rule_and_split():
    assert movielens.binary_label([1, 2, 3, 4, 5]).tolist() == [0, 0, -1, -1, 1]
    tr, te = movielens.train_test_split(1000, 0.7, 0.3, seed=1)
    assert len(tr) == 700 and len(te) == 300 and not set(tr) & set(te)
    tr2, _ = movielens.train_test_split(1000, 0.7, 0.3, seed=1)
    assert tr.tolist() == tr2.tolist()


def test_one_hot_conventions_are_those_of_pandas():
    pd = pytest.importorskip('pandas')
    rows = [l.split('::') for l in USERS.strip().split('\n')] + [['7', 'F', '18', '7', 'x'], ['8', 'M', '50', '20', 'x'],
                                                                   ['9', 'M', '35', '1', 'x']]
    ids, feats, names = movielens.user_features(rows)
    frame = pd.DataFrame({'sex': [r[1] for r in rows], 'occupation': [str(int(r[3])) for r in rows]})
    frame['age_bin'] = pd.cut(pd.Series([int(r[2]) for r in rows]), bins=list(movielens.AGE_BINS),
                              labels=['age_bin_%d' % i for i in range(5)])
    dummies = pd.get_dummies(frame)
    want_names = [c.replace('age_bin_age_bin_', 'age_bin_') for c in dummies.columns]
    assert names == want_names
    np.testing.assert_array_equal(feats, dummies.values.astype(np.float32))


REF_DATA = '/root/reference/data'


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DATA, 'users.dat')), reason="reference data files not here")
def test_real_user_and_movie_tables():
    ids, feats, names = movielens.user_features(movielens.read_dat(os.path.join(REF_DATA, 'users.dat')))
    assert feats.shape == (6040, 2 + 21 + 5) and (feats.sum(1) == 3).all()
    assert names[2:6] == ['occupation_0', 'occupation_1', 'occupation_10', 'occupation_11']
    mids, mfeats, vocab = movielens.movie_features(movielens.read_dat(os.path.join(REF_DATA, 'movies.dat'),
                                                                     encoding='iso-8859-1'))
    assert mfeats.shape == (3883, 18) and vocab[0] == 'Action' and (mfeats.sum(1) >= 1).all()
    rat = movielens.synthetic_ratings(os.path.join(REF_DATA, 'users.dat'), os.path.join(REF_DATA, 'movies.dat'),
                                      20_000, seed=3)
    d = movielens.load_movielens(os.path.join(REF_DATA, 'users.dat'), os.path.join(REF_DATA, 'movies.dat'), rat)
    nu = d['node_features']['user'].shape[0]
    assert d['links'][:, 0].max() == nu - 1 and d['links'][:, 1].min() >= nu
    assert set(d['labels'].tolist()) == {0, 1}
    assert d['edges'].shape[1] % 2 == 0 and (d['edges'][0, :d['edges'].shape[1] // 2] < nu).all()
