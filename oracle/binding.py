"""Oracle (TEST INFRASTRUCTURE ONLY) — dense fp64 restatement of the reference's binding construction,
/root/reference/modules/d3gs/utils/binding_utils.py:105-121 (Mahalanobis test), 199-285 (clip to the max_particles
nearest, uniform weights = softmax of -ones).  O(K N) memory: small cases only.

parity unpinned: the reference implementation needs warp-lang (absent); chi2.ppf comes from scipy as in the reference.
"""
import numpy as np
import torch
from scipy.stats import chi2


def cov6_to_mat(c):
    c = np.asarray(c, dtype=np.float64)
    return np.stack([np.stack([c[:, 0], c[:, 1], c[:, 2]], 1), np.stack([c[:, 1], c[:, 3], c[:, 4]], 1),
                     np.stack([c[:, 2], c[:, 4], c[:, 5]], 1)], 1)


def mahalanobis(means, cov6, particles):
    """(K, N) distances p = d^T inv(cov) d"""
    A = np.linalg.inv(cov6_to_mat(cov6))
    d = np.asarray(particles, np.float64)[None, :, :] - np.asarray(means, np.float64)[:, None, :]
    return np.einsum("kni,kij,knj->kn", d, A, d)


def weight_matrix(means, cov6, particles, confidence=0.95, max_particles=10):
    """Dense (K, N) weight matrix of gaussian_binding_with_clip_v1 + the distances."""
    p = mahalanobis(means, cov6, particles)
    thr = chi2.ppf(confidence, 3)
    W = np.zeros_like(p)
    for k in range(p.shape[0]):
        idx = np.nonzero(p[k] <= thr)[0]
        if idx.size > max_particles:
            idx = idx[np.argsort(p[k, idx], kind="stable")[:max_particles]]
        if idx.size:
            W[k, idx] = 1.0 / idx.size
    return W, p, thr
