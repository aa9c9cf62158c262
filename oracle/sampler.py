"""CPU oracle of the balanced positive / negative sampler (TEST INFRASTRUCTURE ONLY: imported by tests/ -- never by the product path).

Reference: BalancedPositiveNegativeSampler, /root/reference/nerf_rpn/model/utils.py:35-98 -- positives = labels >= 1, negatives =
labels == 0, num_pos = min(#pos, batch * fraction), num_neg = min(#neg, batch - num_pos), each subset drawn with torch.randperm (:79-80).
The reference's draw consumes torch's generator, so WHICH anchors are drawn is not a reproducible contract; what is pinned here is
(a) the counts and class membership rules above and (b) the HIP kernel's own definition of the draw -- the k smallest
(splitmix64(seed, index) >> 32, index) of a class, returned ascending -- restated with numpy uint64 arithmetic for bit-exact parity.
PARITY NOTE: the selection rule (b) has no reference golden vector (the reference's is torch.randperm); uniformity is tested statistically.
"""
import numpy as np

_NEG_SEED = np.uint64(0xD1B54A32D192ED03)


def sample_keys(seed, idx):
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(32)).astype(np.uint64)


def sample_pos_neg(labels, batch, max_pos, seed):
    """labels [T] float -> (pos, neg) ascending int64 index arrays."""
    labels = np.asarray(labels, dtype=np.float32)
    pos = np.nonzero(labels >= 1)[0]
    neg = np.nonzero(labels == 0)[0]
    n_pos = min(pos.size, max_pos)
    n_neg = min(neg.size, batch - n_pos)

    def draw(cand, k, s):
        comp = (sample_keys(s, cand) << np.uint64(32)) | cand.astype(np.uint64)
        return np.sort(cand[np.argsort(comp, kind="stable")[:k]]).astype(np.int64)
    return draw(pos, n_pos, seed), draw(neg, n_neg, int(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ _NEG_SEED))
