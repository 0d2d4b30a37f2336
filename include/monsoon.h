/* monsoon.h -- C ABI of the MI355X batched Stormbound engine (libmonsoon_hip.so).
 *
 * Drop-in boundary for ONE path of dvrp0/Monsoon: the game step / legal-action / observation
 * surface and the heuristic self-play rollouts that score an evolutionary population.  The
 * reference has no FFI (it is pure Python); each entry point below names the Python interface
 * it replaces.  INTEGRATION.md shows the ctypes stubs a maintainer of the reference would add.
 *
 * Conventions: plain C, opaque handle, int status returns (0 = MONSOON_OK), caller-owned HOST
 * buffers unless a name ends in _dev, no callbacks, no global state.  One handle = one HIP
 * device + one stream; a handle is not thread-safe, independent handles are.  Per-game faults
 * (the reference's swallowed Python exceptions) are reported as bytes, never as
 * error returns.  There is no CPU backend: every call fails with MONSOON_ERR_DEVICE when no
 * gfx950 device is usable.
 */
#ifndef MONSOON_H
#define MONSOON_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MONSOON_OK 0
#define MONSOON_ERR_ARG 1       /* bad argument (null pointer, size, illegal action, unsupported card) */
#define MONSOON_ERR_DEVICE 2    /* HIP error; see monsoon_last_error */
#define MONSOON_ERR_STATE 3     /* call order (e.g. step before reset) */

#define MONSOON_NUM_ACTIONS 156 /* enums.py:10-36 */
#define MONSOON_OBS_INTS 540    /* (27,5,4) int32, games/stormbound.py:376-399 */
#define MONSOON_DECK_SIZE 12
#define MONSOON_NUM_FEATURES 10 /* evo/features.py:327-342 */

typedef struct monsoon monsoon_t;

typedef struct {
  int32_t device;          /* HIP device ordinal */
  int32_t max_games;       /* capacity of the batch */
  int32_t lanes_per_game;  /* candidate lanes (successor states stepped at once) per game: 4, 8, 16, 32 or 64; 0 = default.
                              Must be a kernel variant of the build (monsoon_amd/csrc/variants.def), else create fails.
                              (Development knobs read at create: MONSOON_LANES, MONSOON_WPE, and MONSOON_GAMES_PER_WAVE = the
                              kind of hot kernel: 10 the game's record in registers (default of the standard build), 1 in LDS,
                              2 | 4 several games per wavefront -- same results, slower.) */
  int32_t stack_bytes;     /* ignored since the rules core keeps an explicit work stack (kept for ABI compatibility) */
} monsoon_config;

/* One scheduled game of a fitness evaluation (evo/fitness.py:53-59,133-157). */
typedef struct {
  int32_t p1;     /* index into the weight table: row individual, plays FIRST */
  int32_t p2;     /* opponent, plays SECOND */
  uint32_t seed;  /* numpy.random.RandomState(seed) of the game (games/stormbound.py:294) */
  uint32_t deck;  /* index into the deck-pair table passed to monsoon_rollout */
} monsoon_match;

typedef struct {
  uint64_t lookahead_steps;  /* Stormbound.step transitions executed as 1-ply look-ahead */
  uint64_t decisions;        /* committed decisions (each commits one of its look-ahead results) */
  uint64_t games_finished;   /* games that ended with a winner */
  uint64_t faults;           /* games stopped by a fault (reference: swallowed exception -> draw) */
  uint64_t capacity_faults;  /* of those, build-limit faults (must be 0 for a valid run) */
  uint64_t lookahead_capacity_faults;  /* games in which a LOOK-AHEAD hit a build limit: that action scored 0.0 where the
                                          reference computes a score (must be 0 for a valid run) */
} monsoon_stats;

int monsoon_create(const monsoon_config* cfg, monsoon_t** out);
void monsoon_destroy(monsoon_t* h);
const char* monsoon_last_error(monsoon_t* h);   /* h may be NULL: last create() error */
int monsoon_version(void);
/* the hot-kernel variant the handle runs (any pointer may be NULL) */
int monsoon_variant(monsoon_t* h, int32_t* lanes_per_game, int32_t* waves_per_simd);
/* card id string ("u007") -> table index used in deck arrays; -1 if unknown.  card.py:15 */
int monsoon_card_index(const char* card_id);
/* 1 if the card's ability is implemented by this build (decks with other cards are refused) */
int monsoon_card_supported(int card_index);

/* Stormbound.__init__ for n games (games/stormbound.py:293-304, player.py:13-37):
 * seeds[n]; decks[n][2][12] card indices in constructor order; factions[n][2] (enums.py:44-49). */
int monsoon_reset(monsoon_t* h, int32_t n, const uint32_t* seeds, const uint8_t* decks, const uint8_t* factions);

/* Stormbound.legal_actions (games/stormbound.py:528-557) as a 156-bit mask per game: out[n][3]. */
int monsoon_legal_mask(monsoon_t* h, uint64_t* out);

/* Stormbound.step (games/stormbound.py:318-373) for every game; actions[n] must be legal
 * (PASS = 155 is always accepted, as the reference's scripted bot relies on; actions[i] = 255 leaves game i untouched).
 * An illegal entry refuses the whole call (MONSOON_ERR_ARG) before any game is stepped.  reward[n] in {0,1}, done[n], fault[n] (0 = none). */
int monsoon_step(monsoon_t* h, const uint8_t* actions, int8_t* reward, uint8_t* done, uint8_t* fault);

/* Stormbound.expert_action (games/stormbound.py:563-637), the reference's scripted opponent, for every game:
 * out_action[n] (may be PASS while plays remain -- that is how the bot ends a turn); fault[n] (may be NULL)
 * is non-zero where the reference would raise.  Draws from the game's own stream, like the reference. */
int monsoon_expert_action(monsoon_t* h, uint8_t* out_action, uint8_t* fault);

/* Stormbound.get_observation (games/stormbound.py:400-526): out[n][27][5][4] int32.
 * raises[n] = 1 where the reference would raise (int(card) on up01/up02/up03, card.py:46). */
int monsoon_observe(monsoon_t* h, int32_t* out, uint8_t* raises);

/* The same into caller-owned DEVICE memory (a torch-ROCm tensor view (B,27,5,4) int32
 * without a host round trip): out_dev = n*540 int32, raises_dev = n bytes or NULL.  Returns after the
 * handle's stream has finished writing. */
int monsoon_observe_dev(monsoon_t* h, void* out_dev, void* raises_dev);

/* StateFeatures.get_feature_vector (evo/features.py:12-342): out[n][10] float64. */
int monsoon_features(monsoon_t* h, double* out);

/* Stormbound.to_play / have_winner (games/stormbound.py:312-313, 560-561) + bases: out[n][4] =
 * {to_play, have_winner, base_first, base_second}. */
int monsoon_status(monsoon_t* h, int32_t* out);

/* The fault code that stopped each loaded game, else the first build-limit code one of its look-aheads hit (0 = none): out[n].  The reference's exceptions are swallowed by
 * its agent layer (evo/heuristic_agent.py:48-51, evo/fitness.py:170-174,208-210); codes in msb_base.h, >= 16 are
 * limits of this build. */
int monsoon_game_faults(monsoon_t* h, uint8_t* out);

/* Canonical state record of game idx (layout: monsoon_amd/csrc/canon.h), the comparand of the
 * bit-exactness tests.  buf must hold 2048 bytes (the standard and the extended record never need more than 1024). */
int monsoon_state_export(monsoon_t* h, int32_t idx, uint8_t* buf, int32_t* len);

/* copy.deepcopy(game) across the boundary (evo/game_adapter.py:280-287 clone_state): the COMPLETE device state of game
 * idx -- record, bookkeeping row, numpy stream position -- as an opaque blob of monsoon_state_blob_bytes() bytes.
 * monsoon_state_load puts a blob into slot idx of any handle of the same build (idx <= number of loaded games; idx ==
 * that number appends a game, max_games permitting): the clone then continues bit-identically to the original. */
int32_t monsoon_state_blob_bytes(void);
int monsoon_state_save(monsoon_t* h, int32_t idx, uint8_t* buf, int32_t buf_bytes);
int monsoon_state_load(monsoon_t* h, int32_t idx, const uint8_t* buf, int32_t buf_bytes);

/* Diagnostics behind the scenario tests (the reference's own unit tests, test.py:11-147 and the <ID>Test classes, replayed
 * call by call: tests/scenario_lib.py).  monsoon_debug_build puts game idx (idx <= loaded games) into a described
 * state: an int32 stream (layout: monsoon_amd/csrc/scenario.inc) and the position of its numpy stream,
 * RandomState(seed) advanced by stream_pos outputs.  monsoon_debug_op makes ONE call into the engine on that game
 * (Unit.play, activate_ability, deal_damage, destroy, command, respawn, Board.spawn_token_*, to_next_turn,
 * Player.play / discard): *fault = the fault code, log = {card, position} of every ability that ran, in order. */
int monsoon_debug_build(monsoon_t* h, int32_t idx, uint32_t seed, uint32_t stream_pos, const int32_t* state, int32_t n_state, int32_t* fault);
int monsoon_debug_op(monsoon_t* h, int32_t idx, const int32_t* op, int32_t n_op, int32_t* fault, int32_t* log, int32_t log_cap, int32_t* n_log);

/* Diagnostics: the device's numpy-stream draws and score arithmetic on caller-given inputs (numpy.random.RandomState
 * legacy API: player.py:28,49, unit.py:95, cards; np.dot of evo/weights.py:60), for known-answer tests.  kind 0: n raw
 * u32 outputs of RandomState(seed); 1: n random(); 2: randint(0, in[i]) for n int32 bounds; 3: n times
 * shuffle(list(range(12))) (out int32[n][12]); 4: n scores of in = double[n][30] {weights, before, after}.  n <= 4096.
 * Re-seeds the stream buffers of game slot 0. */
int monsoon_debug_kat(monsoon_t* h, int32_t kind, uint32_t seed, int32_t n, const void* in, void* out);

/* Debugging aid: the raw HBM record of game idx (monsoon_amd/csrc/state.h layout); buf must hold 4096 bytes. */
int monsoon_debug_raw(monsoon_t* h, int32_t idx, uint8_t* buf, int32_t* len);

/* FNV-1a 64 of the canonical record of every game: out[n].  Lets a caller compare a whole batch
 * against a replay without exporting 65 536 records one by one. */
int monsoon_state_hash(monsoon_t* h, uint64_t* out);

/* HeuristicAgent.select_action + adapter.apply_action for every live game
 * (evo/heuristic_agent.py:53-80, evo/game_adapter.py:320-325): 1-ply look-ahead over all legal
 * actions, score, first-max argmax, commit.  weights[n][2][10]: the weight vector of the FIRST
 * and SECOND player of each game.  out_action[n] (255 = game already over), out_score[n]
 * (best score), out_scores[n][156] (NaN = illegal; may be NULL). */
int monsoon_decide(monsoon_t* h, const double* weights, uint8_t* out_action, double* out_score, double* out_scores);

/* FitnessEvaluator rollouts (evo/fitness.py:123-228 with the corrected loop of SURVEY.md §8c):
 * plays n_matches games to the end (winner, fault, or max_turns decisions), at most
 * cfg.max_games at a time.  weights[n_individuals][10]; deck_pairs[n_decks][2][12];
 * out_counts[n_individuals][3] = {wins, draws, games} of each individual as p1, ACCUMULATED
 * into the caller's buffer; out_results[n_matches] (may be NULL): -1 draw, 0 p1, 1 p2;
 * out_steps[n_matches] (may be NULL): decisions played. */
int monsoon_rollout(monsoon_t* h, const double* weights, int32_t n_individuals, const monsoon_match* matches,
                    int32_t n_matches, const uint8_t* deck_pairs, int32_t n_decks, int32_t max_turns,
                    int32_t* out_counts, int8_t* out_results, int32_t* out_steps);
/* Fault code of every game of the last completed monsoon_rollout, out[n_matches] (same meaning as monsoon_game_faults:
 * the fault that stopped the game, else the first capacity code one of its look-aheads hit).  Codes >= 16 are limits
 * of this build's record, not reference behaviour (the exception the reference swallows at evo/fitness.py:170-174,
 * 208-210 is code 1): such games are replayed on a build with a larger record -- libmonsoon_hip_ext.so ->
 * libmonsoon_hip_big.so, as monsoon_amd/fitness.py does -- and their rows of the result replaced. */
int monsoon_rollout_faults(monsoon_t* h, uint8_t* out, int32_t n_matches);

/* Per-game decks of configuration C5 on the device: for every seed, numpy.random.RandomState(seed).choice(pool, 12,
 * replace=False) twice -> out_pairs[n][2][12] (card indices taken from pool[pool_n], 12 <= pool_n <= 128).  The caller
 * passes the pre-stream seeds (game seed ^ 0x9E3779B9).  Host buffers in and out; needs no loaded games.
 * Replaces the per-game Python draw of games/evolutionary_stormbound.py:52 + utils.py:26-119 for this configuration
 * (150 us per game in numpy, 79 s for one C5 generation). */
int monsoon_draw_decks(monsoon_t* h, const uint32_t* seeds, int32_t n, const uint8_t* pool, int32_t pool_n, uint8_t* out_pairs);

/* GA operators on the device.  The host GA (monsoon_amd/population.py, as the reference's
 * evo/population.py) stays the default and the bit-exact path; these are the same operators over the same numpy stream
 * for a driver that wants the population to stay next to the rollouts.
 *
 * monsoon_np_state = numpy.random.RandomState.get_state(legacy=False) of the GLOBAL stream the reference draws from
 * (np.random.*): the mt19937 key, its position, and the cached second normal of legacy_gauss; in/out.
 *
 * monsoon_ga_offspring: Population.generate_offspring, evo/population.py:75-89 with WeightVector.copy / mutate,
 * evo/weights.py:12-40: lambda offspring of parents[mu][dim] (weights, sigmas).  out_parent[lambda] = the parent drawn for
 * each child, out_tries[lambda] = polar-method rounds consumed up to and including that child (both optional).  Every
 * draw and every accept / reject decision is numpy's own, so the parents and the returned stream state (key, position,
 * has_gauss) are bit-identical to the host's; exp / log / sqrt are the device library's, within 1 ulp of the host's,
 * so weights and sigmas agree with the host's to a few ulp (tests/test_gpu_parity.py).
 *
 * monsoon_ga_select: the order of select_from_combined, evo/population.py:91-103: indices of all n individuals by
 * descending fitness, equal fitness in original order (Python's stable sort with reverse=True). */
typedef struct {
  uint32_t key[624];
  int32_t pos;        /* 0..624 (624 = the next draw regenerates the key) */
  int32_t has_gauss;
  double gauss;
} monsoon_np_state;
int monsoon_ga_offspring(monsoon_t* h, monsoon_np_state* st, const double* parents_w, const double* parents_s, int32_t mu, int32_t dim,
                         int32_t lambda, double tau, double tau_prime, double min_sigma, double* out_w, double* out_s, int32_t* out_parent,
                         int64_t* out_tries);
int monsoon_ga_select(monsoon_t* h, const double* fitness, int32_t n, int32_t* out_order);

/* Diagnostics: 192 raw counter words (words 0-4 back monsoon_get_stats; a profiling build (-DMSB_PROF=1,
 * scripts only) adds k_decide phase cycles at 8..15, per-function cycles / calls at 32..63 / 64..95, last-launch
 * occupancy at 96..101 and call entry / exit cycles at 128..159 / 160..191; the rest is zero).
 * No reference counterpart. */
int monsoon_debug_counters(monsoon_t* h, unsigned long long* out192);

/* Device-resident variant used by bench.py: one decision round over the games already loaded by
 * monsoon_reset, weights taken from a table uploaded once.  Nothing crosses PCIe. */
int monsoon_upload_weights(monsoon_t* h, const double* weights, int32_t n_individuals);
int monsoon_assign_players(monsoon_t* h, const int32_t* p1, const int32_t* p2);   /* [n] indices */
int monsoon_decide_round_dev(monsoon_t* h);   /* asynchronous on the handle's stream */
/* `rounds` decisions of every loaded game in one launch (evo/fitness.py:193-211, a slice of the _play_game loop): a
 * game's record stays on chip from its first to its last decision of the call.  Asynchronous like the call above. */
int monsoon_play_rounds_dev(monsoon_t* h, int32_t rounds);
int monsoon_sync(monsoon_t* h);

int monsoon_get_stats(monsoon_t* h, monsoon_stats* out);
int monsoon_reset_stats(monsoon_t* h);
/* HIP-event timing of the decide kernel since the last reset_stats: total ms and launch count. */
int monsoon_kernel_time(monsoon_t* h, double* total_ms, int64_t* launches);
/* The stream the handle launches on (hipStream_t) so a caller can order its own work; NULL-safe.  Every handle has a
 * stream of its own (one handle = one device + one stream; independent handles are independent).  MONSOON_OWN_STREAM=0
 * in the environment puts all handles on the device's default stream (returned as NULL), as round 2 had to. */
void* monsoon_stream(monsoon_t* h);

#ifdef __cplusplus
}
#endif
#endif /* MONSOON_H */
