"""WeightVector -- the GA individual (mirror of evo/weights.py:12-123): 10 weights in [0,1] and
self-adaptive mutation strengths.  All randomness comes from the global numpy stream in the same
call order as the reference, so a seeded run draws the same individuals."""
import numpy as np


class WeightVector:
    def __init__(self, size):
        self.weights = np.random.uniform(0, 1, size)        # evo/weights.py:16
        self.sigmas = np.full(size, 0.1)
        self.size = size

    def mutate(self, tau, tau_prime, min_sigma):
        """sigma_i' = max(sigma_i * exp(tau'*N + tau*N_i), eps); w_i' = clip(w_i + N(0, sigma_i'), 0, 1)
        (evo/weights.py:20-40; draw order: global normal, per-gene normals, per-gene steps)."""
        g = np.random.normal(0, 1)
        per_gene = np.random.normal(0, 1, len(self.sigmas))
        self.sigmas = np.maximum(self.sigmas * np.exp(tau_prime * g + tau * per_gene), min_sigma)
        self.weights = np.clip(self.weights + np.random.normal(0, self.sigmas), 0, 1)

    def copy(self):
        other = WeightVector(len(self.weights))   # consumes `size` uniforms, as the reference's copy does
        other.weights = self.weights.copy()
        other.sigmas = self.sigmas.copy()
        other.size = self.size
        return other

    def distance_to(self, other):
        if self.size != other.size:
            raise ValueError("Cannot compute distance between vectors of different sizes")
        return np.linalg.norm(self.weights - other.weights)

    def dot_product(self, features):
        if len(features) != self.size:
            raise ValueError(f"Feature vector size {len(features)} doesn't match weight vector size {self.size}")
        return np.dot(self.weights, features)

    def get_weights(self):
        return self.weights.copy()

    def get_sigmas(self):
        return self.sigmas.copy()

    def set_weights(self, weights):
        if len(weights) != self.size:
            raise ValueError(f"Weight array size {len(weights)} doesn't match expected size {self.size}")
        self.weights = np.clip(weights, 0, 1)

    def set_sigmas(self, sigmas):
        if len(sigmas) != self.size:
            raise ValueError(f"Sigma array size {len(sigmas)} doesn't match expected size {self.size}")
        self.sigmas = np.maximum(sigmas, 1e-10)

    def normalize_weights(self):
        s = np.sum(self.weights)
        if s > 0:
            self.weights = self.weights / s

    def __repr__(self):
        return f"WeightVector(size={self.size}, weights={self.weights}, sigmas={self.sigmas})"

    @classmethod
    def from_arrays(cls, weights, sigmas=None):
        v = cls(len(weights))
        v.set_weights(weights)
        if sigmas is not None:
            v.set_sigmas(sigmas)
        return v

    @classmethod
    def zeros(cls, size):
        v = cls(size)
        v.weights = np.zeros(size)
        return v

    @classmethod
    def ones(cls, size):
        v = cls(size)
        v.weights = np.ones(size)
        return v

    @classmethod
    def random_uniform(cls, size, low=0.0, high=1.0):
        v = cls(size)
        v.weights = np.random.uniform(low, high, size)
        return v
