"""Population -- (mu + lambda) evolution strategy bookkeeping (mirror of evo/population.py).

Stays on the host, as in the reference.  The numpy global-stream call order of
initialize_population / generate_offspring / select_from_combined follows the reference exactly
(evo/population.py:28-69, 75-89, 91-176), so `config.seed` reproduces its individuals; this is
pinned by tests/golden/population_seed42.npz.
"""
import pickle

import numpy as np

from .weights import WeightVector


class Population:
    def __init__(self, config):
        self.config = config
        self.individuals = []
        self.fitness_scores = []
        self.generation = 0
        if config.seed is not None:
            np.random.seed(config.seed)

    # evo/population.py:28-69 -- three initialisation groups + per-individual sigma spread
    def initialize_population(self, feature_count):
        mu = self.config.mu
        self.individuals = []
        for i in range(mu):
            ind = WeightVector(feature_count)
            if i < mu // 3:
                w = np.random.uniform(0.2, 0.8, feature_count)
            elif i < 2 * mu // 3:
                w = np.random.choice([0.0, 1.0], feature_count, p=[0.3, 0.7])
                w = np.clip(w + np.random.normal(0, 0.1, feature_count), 0, 1)
            else:
                w = np.random.uniform(0.0, 1.0, feature_count)
            ind.set_weights(w)
            spread = np.random.uniform(0.5, 2.0)
            sig = np.full(feature_count, self.config.initial_sigma * spread)
            ind.set_sigmas(sig * np.random.uniform(0.8, 1.2, feature_count))
            self.individuals.append(ind)
        self.fitness_scores = [0.0] * mu

    def get_parents(self):
        return self.individuals[:self.config.mu]

    # evo/population.py:75-89
    def generate_offspring(self, engine=None):
        """engine: a BatchEngine -> the device operator (monsoon_ga_offspring) over the same global numpy stream: the same
        parents, the same stream position afterwards, weights / sigmas within a few ulp (device exp / log).  Default and
        bit-exact path: the host loop below."""
        if engine is not None:
            pw = np.stack([p.weights for p in self.individuals[:self.config.mu]])
            ps = np.stack([p.sigmas for p in self.individuals[:self.config.mu]])
            ow, osg, _, _, state = engine.ga_offspring(np.random.get_state(), pw, ps, self.config.lambda_, self.config.tau, self.config.tau_prime,
                                                       self.config.min_sigma)
            np.random.set_state(state)
            children = []
            for k in range(self.config.lambda_):   # no constructor call: WeightVector(size) would draw from the stream
                v = WeightVector.__new__(WeightVector)
                v.weights, v.sigmas, v.size = ow[k].copy(), osg[k].copy(), ow.shape[1]
                children.append(v)
            return children
        children = []
        for _ in range(self.config.lambda_):
            parent = self.individuals[np.random.randint(0, self.config.mu)]
            child = parent.copy()
            child.mutate(self.config.tau, self.config.tau_prime, self.config.min_sigma)
            children.append(child)
        return children

    # evo/population.py:91-176
    def select_from_combined(self, all_individuals, fitness_scores, order=None):
        """order: the ranking computed on the device (BatchEngine.ga_select: descending fitness, ties in original order) --
        the same permutation Python's stable sort gives; None = sort here."""
        if len(all_individuals) != len(fitness_scores):
            raise ValueError(f"Individuals ({len(all_individuals)}) must match fitness scores ({len(fitness_scores)})")
        if order is not None:
            ranked = [(fitness_scores[int(i)], all_individuals[int(i)]) for i in order[:self.config.mu]]
        else:
            ranked = sorted(zip(fitness_scores, all_individuals), key=lambda p: p[0], reverse=True)[:self.config.mu]
        self.fitness_scores = [p[0] for p in ranked]
        self.individuals = [p[1] for p in ranked]
        self.generation += 1
        cfg = self.config
        sigmas = np.array([ind.get_sigmas() for ind in self.individuals])
        if np.mean(sigmas) < cfg.min_sigma * 10:
            # mutation strengths collapsed: redraw them around initial_sigma
            for ind in self.individuals:
                ind.set_sigmas(np.random.uniform(cfg.initial_sigma * 0.5, cfg.initial_sigma * 1.5, len(ind.get_sigmas())))
        std = np.std(self.fitness_scores)
        if std < 1e-3 and std == 0.0 and len(set(self.fitness_scores)) == 1:
            # every individual scored the same: strongly mutate a random half
            k = max(1, len(self.individuals) // 2)
            for idx in np.random.choice(len(self.individuals), k, replace=False):
                ind = self.individuals[idx]
                keep = ind.get_sigmas().copy()
                ind.set_sigmas(keep * 5.0)
                for _ in range(3):
                    ind.mutate(cfg.tau * 2, cfg.tau_prime * 2, cfg.min_sigma)
                ind.set_sigmas(np.maximum(keep, cfg.initial_sigma * 0.5))

    def get_best_individual(self):
        if not self.fitness_scores:
            raise ValueError("No fitness scores available")
        i = int(np.argmax(self.fitness_scores))
        return self.individuals[i], self.fitness_scores[i]

    def get_population_stats(self):
        if not self.fitness_scores:
            return {"error": "No fitness scores available"}
        f = np.array(self.fitness_scores)
        w = np.array([ind.get_weights() for ind in self.individuals])
        s = np.array([ind.get_sigmas() for ind in self.individuals])
        return {"generation": self.generation, "population_size": len(self.individuals), "best_fitness": float(f.max()),
                "worst_fitness": float(f.min()), "mean_fitness": float(f.mean()), "std_fitness": float(f.std()),
                "diversity": float(np.mean(np.std(w, axis=0))), "avg_mutation_strength": float(np.mean(s))}

    # evo/population.py:253-279
    def should_terminate(self):
        cfg = self.config
        if self.generation >= cfg.generations:
            return True
        if self.generation > max(10, cfg.generations // 10) and len(self.fitness_scores) > 1:
            std, mean = np.std(self.fitness_scores), np.mean(self.fitness_scores)
            identical = len(set(np.round(self.fitness_scores, 10))) == 1
            if std < 1e-10 and abs(mean) > 1e-3 and identical and self.generation > cfg.generations // 2:
                return True
        return False

    # evo/population.py:281-310: pickle of {generation, individuals, fitness_scores, config}
    def save_population(self, filepath):
        with open(filepath, "wb") as f:
            pickle.dump({"generation": self.generation, "individuals": self.individuals,
                         "fitness_scores": self.fitness_scores, "config": self.config}, f)

    def load_population(self, filepath):
        with open(filepath, "rb") as f:   # files written by save_population above
            data = pickle.load(f)
        self.generation = data["generation"]
        self.individuals = data["individuals"]
        self.fitness_scores = data["fitness_scores"]
        self.config = data["config"]

    def save_reference_checkpoint(self, filepath):
        """The same checkpoint in the REFERENCE's own class names: a pickle that evo/population.py:295-310
        (Population.load_population) of an unmodified reference loads -- {generation, individuals, fitness_scores, config}
        with individuals of class evo.weights.WeightVector ({weights, sigmas, size}) and a config of class
        evo.config.EvolutionaryConfig (the reference's field names; this build's extra fields ride along as attributes).
        Nothing of the reference is imported: stand-in classes carrying the reference's module and class names exist only
        while the pickle is written (pickle records names, not code)."""
        import sys
        import types
        mods = {}

        def stand_in(module, name):
            m = mods.setdefault(module, types.ModuleType(module))
            cls = type(name, (object,), {"__module__": module})
            setattr(m, name, cls)
            return cls
        WV, CFG = stand_in("evo.weights", "WeightVector"), stand_in("evo.config", "EvolutionaryConfig")
        mods.setdefault("evo", types.ModuleType("evo"))

        def conv(v):
            o = WV.__new__(WV)
            o.__dict__.update({"weights": np.array(v.weights, dtype=np.float64), "sigmas": np.array(v.sigmas, dtype=np.float64), "size": int(v.size)})
            return o
        cfg = CFG.__new__(CFG)
        cfg.__dict__.update(self.config.to_dict())
        data = {"generation": self.generation, "individuals": [conv(v) for v in self.individuals],
                "fitness_scores": list(self.fitness_scores), "config": cfg}
        saved = {k: sys.modules.get(k) for k in mods}
        sys.modules.update(mods)
        try:
            with open(filepath, "wb") as f:
                pickle.dump(data, f, protocol=4)
        finally:
            for k, v in saved.items():
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v

    def __len__(self):
        return len(self.individuals)
