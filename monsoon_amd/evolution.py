"""EvolutionEngine -- the GA driver loop (mirror of evo/evolution.py:18-276).  Stays Python.

initialize(); run(): while not population.should_terminate(): parents (+ offspring after
generation 0) -> FitnessEvaluator.evaluate_population -> assign / (mu+lambda) selection ->
training_log.csv row -> periodic checkpoint; then results_summary.txt.
"""
import os
import time
from datetime import datetime

from .fitness import FitnessEvaluator
from .population import Population

FEATURE_COUNT = 10   # StateFeatures.get_feature_count(), evo/features.py:345-347
FEATURE_NAMES = ["mana_efficiency", "health_advantage", "board_control", "front_line_advantage", "total_strength",
                 "unit_count", "structure_count", "threatened_base", "protection_value", "hand_quality"]


class EvolutionEngine:
    def __init__(self, config, deck_config=None, rollout_fn=None):
        self.config = config
        self.deck_config = deck_config
        self.population = None
        self.fitness_evaluator = None
        self.start_time = None
        self.results_dir = config.results_dir
        self._rollout_fn = rollout_fn
        os.makedirs(self.results_dir, exist_ok=True)

    def initialize(self):
        self.population = Population(self.config)
        self.population.initialize_population(FEATURE_COUNT)
        self.fitness_evaluator = FitnessEvaluator(self.config, self.deck_config, rollout_fn=self._rollout_fn)
        self.fitness_evaluator.warm_up()   # device buffers are allocated here, not inside generation 0's evaluation

    def run(self):
        if self.population is None or self.fitness_evaluator is None:
            raise ValueError("Engine not initialized. Call initialize() first.")
        self.start_time = time.time()
        pop = self.population
        while not pop.should_terminate():
            t0 = time.time()
            # config.ga_on_device: offspring and the selection order through monsoon_ga_* on the evaluator's engine
            ga = self.fitness_evaluator._engine(0) if (self.config.ga_on_device and self._rollout_fn is None) else None
            everyone = pop.get_parents()
            if pop.generation > 0:
                everyone = everyone + pop.generate_offspring(engine=ga)
            scores = self.fitness_evaluator.evaluate_population(everyone, pop.generation)
            if pop.generation == 0:
                pop.fitness_scores = scores
                pop.generation += 1
            else:
                pop.select_from_combined(everyone, scores, order=ga.ga_select(scores) if ga is not None else None)
            self._log_generation(time.time() - t0)
            if pop.generation % self.config.checkpoint_interval == 0:
                self._save_checkpoint()
        return self._finalize_training(time.time() - self.start_time)

    def _log_generation(self, generation_time):
        stats = self.population.get_population_stats()
        ev = self.fitness_evaluator.get_stats()
        print(f"gen {stats['generation']}: best {stats['best_fitness']:.4f} mean {stats['mean_fitness']:.4f} "
              f"std {stats['std_fitness']:.4f} games/s {ev['games_per_second']:.1f} env-steps/s {ev['env_steps_per_second']:.3g}")
        if not self.config.save_logs:
            return
        path = os.path.join(self.results_dir, "training_log.csv")
        if not os.path.exists(path):
            with open(path, "w") as f:   # the reference's columns + env_steps_per_sec
                f.write("generation,time,best_fitness,mean_fitness,std_fitness,diversity,avg_sigma,games_per_sec,env_steps_per_sec\n")
        with open(path, "a") as f:
            f.write(f"{stats['generation']},{generation_time:.2f},{stats['best_fitness']:.6f},{stats['mean_fitness']:.6f},"
                    f"{stats['std_fitness']:.6f},{stats['diversity']:.6f},{stats['avg_mutation_strength']:.6f},"
                    f"{ev['games_per_second']:.1f},{ev['env_steps_per_second']:.1f}\n")

    def _save_checkpoint(self):
        stamp = datetime.now().strftime("%Y%m%d_%H%M%S")
        self.population.save_population(os.path.join(self.results_dir, f"checkpoint_gen{self.population.generation}_{stamp}.pkl"))

    def _finalize_training(self, total_time):
        stats = self.population.get_population_stats()
        best, best_fitness = self.population.get_best_individual()
        final = os.path.join(self.results_dir, "final_population.pkl")
        self.population.save_population(final)
        results = {"total_time": total_time, "generations": self.population.generation, "final_stats": stats,
                   "best_fitness": best_fitness, "best_weights": best.get_weights().tolist(),
                   "evaluation_stats": self.fitness_evaluator.get_stats(), "final_population_file": final}
        with open(os.path.join(self.results_dir, "results_summary.txt"), "w") as f:
            f.write(f"generations {self.population.generation}\nbest_fitness {best_fitness:.6f}\n")
            for name, w in zip(FEATURE_NAMES, best.get_weights()):
                f.write(f"{name} {w:.6f}\n")
        return results

    def load_checkpoint(self, checkpoint_file):
        if self.population is None:
            self.population = Population(self.config)
        self.population.load_population(checkpoint_file)
        if self.fitness_evaluator is None:
            self.fitness_evaluator = FitnessEvaluator(self.config, self.deck_config, rollout_fn=self._rollout_fn)
