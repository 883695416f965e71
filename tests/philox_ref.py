"""TEST INFRASTRUCTURE ONLY: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC 2011) in numpy, pinned on the
known-answer vectors of the authors' Random123 distribution (kat_vectors), and the draw layout of the fused training iteration's generator
(csrc/ngp_march.hip: train_draw / u01) restated on top of it.  The HIP kernel is compared with this, bit for bit (tests/test_gpu_ngp_trainer.py)."""
import numpy as np

KAT = (  # (counter, key) -> output, Random123 kat_vectors, philox4x32 10 rounds
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
)


def philox4x32_10(ctr, key):
    """ctr: four uint32 arrays (or ints), key: two -> four uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) for x in ctr]
    k = [np.asarray(x, dtype=np.uint64) for x in key]
    m0, m1, mask, s = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xffffffff), np.uint64(32)
    for _ in range(10):
        p0, p1 = m0 * c[0], m1 * c[2]
        c = [((p1 >> s) ^ c[1] ^ k[0]) & mask, p1 & mask, ((p0 >> s) ^ c[3] ^ k[1]) & mask, p0 & mask]
        k = [(k[0] + np.uint64(0x9E3779B9)) & mask, (k[1] + np.uint64(0xBB67AE85)) & mask]
    return [x.astype(np.uint32) for x in c]


def u01(w):
    """24 random bits -> f32 in [0, 1) (csrc/ngp_march.hip: u01)."""
    return (np.asarray(w, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def train_draw(seed: int, iteration: int, stream: int, index):
    """Four words per (seed, iteration, stream, global ray index): counter = (index lo, index hi, iteration lo, iteration hi ^ stream << 31), key = seed."""
    index = np.asarray(index, dtype=np.uint64)
    lo32 = lambda v: np.asarray(v, dtype=np.uint64) & np.uint64(0xffffffff)
    hi32 = lambda v: np.asarray(v, dtype=np.uint64) >> np.uint64(32)
    it = np.uint64(iteration)
    return philox4x32_10((lo32(index), hi32(index), np.broadcast_to(lo32(it), index.shape), np.broadcast_to(hi32(it) ^ np.uint64((stream << 31) & 0xffffffff), index.shape)),
                         (lo32(np.uint64(seed)), hi32(np.uint64(seed))))


def background(seed: int, iteration: int) -> np.ndarray:
    w = train_draw(seed, iteration, 1, np.zeros(1, dtype=np.uint64))
    return np.array([u01(w[0])[0], u01(w[1])[0], u01(w[2])[0]], dtype=np.float32)
