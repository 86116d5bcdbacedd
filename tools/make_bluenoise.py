"""Generates akari_render_amd/data/bluenoise_128x128x48_u16.bin: 48 toroidal 128 x 128 blue-noise dither arrays (every rank
0 .. 16383 exactly once, scaled to u16), void-and-cluster (Ulichney 1993) with a Gaussian energy of sigma = 1.9.

The reference's PMJ02BN sampler (crates/akari_render/src/sampler/mod.rs:329-700) offsets its point sets per pixel with such
textures (akari_data::bluenoise, a copy of pbrt-v4's tables); that data file is absent from the reference tree here
(.MISSING_LARGE_BLOBS), so these arrays are REGENERATED: same shape, same role, different values -- images rendered with
the pmj02bn sampler are therefore not comparable bit-for-bit with the reference's.

    python tools/make_bluenoise.py [n_textures=48] [seed=2024]
"""
import os
import sys
import time

import numpy as np

N = 128
SIGMA = 1.9


def kernel():
    d = np.minimum(np.arange(N), N - np.arange(N)).astype(np.float64)
    g = np.exp(-(d * d) / (2 * SIGMA * SIGMA))
    return np.outer(g, g)


K = kernel()


def shifted(y, x):
    return np.roll(np.roll(K, y, axis=0), x, axis=1)


def void_and_cluster(rng):
    n_pix = N * N
    n_init = n_pix // 10
    pat = np.zeros((N, N), dtype=bool)
    idx = rng.choice(n_pix, n_init, replace=False)
    pat.flat[idx] = True
    energy = np.real(np.fft.ifft2(np.fft.fft2(pat.astype(np.float64)) * np.fft.fft2(K)))
    # phase 0: relax the initial pattern (move the tightest cluster into the largest void until they coincide)
    while True:
        c = np.argmax(np.where(pat, energy, -np.inf))
        cy, cx = divmod(c, N)
        pat[cy, cx] = False
        energy -= shifted(cy, cx)
        v = np.argmin(np.where(pat, np.inf, energy))
        vy, vx = divmod(v, N)
        pat[vy, vx] = True
        energy += shifted(vy, vx)
        if v == c:
            break
    rank = np.zeros((N, N), dtype=np.int64)
    # phase 1: remove tightest clusters, ranks n_init-1 .. 0
    p, e = pat.copy(), energy.copy()
    for r in range(n_init - 1, -1, -1):
        c = np.argmax(np.where(p, e, -np.inf))
        cy, cx = divmod(c, N)
        p[cy, cx] = False
        e -= shifted(cy, cx)
        rank[cy, cx] = r
    # phase 2 + 3: fill the largest voids, ranks n_init .. n_pix-1 (the minority / majority switch of the original paper
    # only changes which energy is tracked; with a symmetric kernel the largest void of the ones is the same pixel)
    p, e = pat.copy(), energy.copy()
    for r in range(n_init, n_pix):
        v = np.argmin(np.where(p, np.inf, e))
        vy, vx = divmod(v, N)
        p[vy, vx] = True
        e += shifted(vy, vx)
        rank[vy, vx] = r
    return rank


def main():
    n_tex = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
    out = np.zeros((n_tex, N, N), dtype=np.uint16)
    t0 = time.time()
    for t in range(n_tex):
        rank = void_and_cluster(np.random.default_rng(seed + t))
        assert np.array_equal(np.sort(rank.ravel()), np.arange(N * N))
        out[t] = ((rank.astype(np.float64) + 0.5) * (65536.0 / (N * N))).astype(np.uint16)
        print(f"texture {t + 1}/{n_tex}  ({time.time() - t0:.0f} s)", flush=True)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "akari_render_amd", "data", f"bluenoise_128x128x{n_tex}_u16.bin")
    out.tofile(path)
    print("wrote", path, out.nbytes, "bytes")


if __name__ == "__main__":
    main()
