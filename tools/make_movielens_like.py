"""Write users.dat and movies.dat of MovieLens-1M shape (6040 users, 3883 movies,
18 genres, 21 occupations, 7 age codes) filled with random values, for boxes that
do not hold the real files.  ``python tools/make_movielens_like.py OUT_DIR``."""
import os
import sys

import numpy as np

GENRES = ["Action", "Adventure", "Animation", "Children's", "Comedy", "Crime", "Documentary", "Drama", "Fantasy",
          "Film-Noir", "Horror", "Musical", "Mystery", "Romance", "Sci-Fi", "Thriller", "War", "Western"]
AGES = [1, 18, 25, 35, 45, 50, 56]


def main(out, n_users=6040, n_movies=3883, seed=0):
    os.makedirs(out, exist_ok=True)
    rng = np.random.default_rng(seed)
    with open(os.path.join(out, 'users.dat'), 'w') as fh:
        for u in range(1, n_users + 1):
            fh.write("%d::%s::%d::%d::%05d\n" % (u, 'FM'[rng.integers(2)], AGES[rng.integers(7)], rng.integers(21),
                                                 rng.integers(99999)))
    with open(os.path.join(out, 'movies.dat'), 'w', encoding='iso-8859-1') as fh:
        for m in range(1, n_movies + 1):
            k = rng.integers(1, 4)
            fh.write("%d::Film %d (1999)::%s\n" % (m, m, '|'.join(rng.choice(GENRES, k, replace=False))))


if __name__ == '__main__':
    main(sys.argv[1])
