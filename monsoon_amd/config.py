"""EvolutionaryConfig -- host-side mirror of the reference's GA configuration (evo/config.py:10-146).

Same field names, defaults, validation and nested-JSON layout, so a config.json written for the
reference loads here unchanged.  Fields below the marker are additions of this build (device
rollouts); the reference ignores unknown keys only if they are stripped, see INTEGRATION.md.
"""
import dataclasses
import json
from typing import Optional


@dataclasses.dataclass
class EvolutionaryConfig:
    mu: int = 10
    lambda_: int = 10
    generations: int = 100
    games_per_pairing: int = 20
    deck_configs: int = 3
    tau: float = 0.1
    tau_prime: float = 0.01
    min_sigma: float = 1e-5
    initial_sigma: float = 0.1
    sigma_reset_threshold: float = 1e-4
    sigma_boost_factor: float = 2.0
    fitness_stagnation_gens: int = 10
    max_turns: int = 100
    num_workers: int = 128
    timeout_seconds: int = 30
    checkpoint_interval: int = 10
    save_best_n: int = 5
    log_level: str = "INFO"
    save_logs: bool = True
    save_generation_details: bool = True
    track_weight_evolution: bool = True
    results_dir: str = "results/evolutionary2"
    min_generations: int = 50
    fitness_plateau_threshold: float = 0.001
    plateau_generations: int = 25
    seed: Optional[int] = None
    # ---- additions of this build ------------------------------------------------------------
    mode: str = "rollout"            # "rollout" = corrected loop; "as_written" = reference bug-for-bug
    schedule: str = "round_robin"    # "round_robin" (evo/fitness.py:53-59) or "ring" 
    games_per_individual: int = 64   # ring schedule only
    deck: str = "N12M"               # key of monsoon_amd.cards.DECKS, both sides; "random109" = per-game decks of configuration C5 (cards.RANDOM_DECK)
    max_concurrent_games: int = 65536
    lanes_per_game: int = 0          # hot-kernel variant: candidate lanes per game (0 = build default)
    concurrent_tiers: bool = True    # a schedule with games on both records: the two sub-schedules from two host threads (two handles, two streams)
    ga_on_device: bool = False       # offspring + selection order through monsoon_ga_* (same numpy stream; results within a few ulp of the host's)

    # nested-JSON sections of the reference's configs/config.json -> flat fields
    _SECTIONS = {
        "population": ("mu", "lambda_"),
        "evolution": ("generations",),
        "evaluation": ("games_per_pairing", "deck_configs"),
        "mutation": ("tau", "tau_prime", "min_sigma", "initial_sigma", "sigma_reset_threshold", "sigma_boost_factor",
                     "fitness_stagnation_gens"),
        "simulation": ("max_turns", "num_workers", "timeout_seconds"),
        "checkpointing": ("checkpoint_interval", "save_best_n", "results_dir"),
        "logging": ("log_level", "save_generation_details", "track_weight_evolution"),
        "convergence_criteria": ("min_generations", "fitness_plateau_threshold", "plateau_generations"),
    }

    def __post_init__(self):
        for name, ok, msg in (
            ("mu", self.mu > 0, "Parent population size (mu) must be positive"),
            ("lambda_", self.lambda_ > 0, "Offspring size (lambda_) must be positive"),
            ("generations", self.generations > 0, "Generations must be positive"),
            ("games_per_pairing", self.games_per_pairing > 0, "Games per pairing must be positive"),
            ("tau", self.tau > 0 and self.tau_prime > 0, "Mutation parameters (tau, tau_prime) must be positive"),
            ("min_sigma", self.min_sigma > 0, "Minimum sigma must be positive"),
            ("num_workers", self.num_workers > 0, "Number of workers must be positive"),
        ):
            if not ok:
                raise ValueError(msg)
        if self.mode not in ("rollout", "as_written"):
            raise ValueError("mode must be 'rollout' or 'as_written'")
        if self.schedule not in ("round_robin", "ring"):
            raise ValueError("schedule must be 'round_robin' or 'ring'")

    @classmethod
    def from_json(cls, json_path):
        with open(json_path) as f:
            data = json.load(f)
        flat = {}
        for section, keys in cls._SECTIONS.items():
            if section in data:
                for k in keys:
                    flat[k] = data[section][k]   # KeyError on a missing key, like the reference
        if "device" in data:   # optional section of this build
            flat.update(data["device"])
        return cls(**flat)

    @classmethod
    def from_dict(cls, d):
        return cls(**d)

    def to_dict(self):
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self)}

    def save_json(self, path):
        with open(path, "w") as f:
            json.dump(self.to_dict(), f, indent=2)
