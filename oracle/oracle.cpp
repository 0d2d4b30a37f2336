// oracle/oracle.cpp -- CPU replay oracle.  TEST INFRASTRUCTURE, NOT PRODUCT.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this library; the
// product (monsoon_amd/, libmonsoon_hip.so) never links, imports or calls it and has no CPU path.
//
// What it is: the host compilation (g++) of the rules core in monsoon_amd/csrc/ -- rules.h,
// abilities.inc, observe.inc, mt19937.h -- which restates the reference engine function by
// function (each cites its reference file:line), run scalar, one game at a time, with the
// state in a plain byte array.  It is pinned to the Python reference by the golden vectors under
// tests/golden/ (generated in the build container by oracle/pyref/gen_golden.py from
// /root/reference itself) and by the live differential fuzzer oracle/pyref/difffuzz.py.
// The GPU tests compare the HIP kernels against BOTH this oracle and those golden vectors, so a
// bug shared by the two builds of the rules core still has to get past the reference's own outputs.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>

// Two builds of this file (oracle/Makefile):
//   liboracle*.so        the ORACLE: oracle/recursive/ -- the rules core written the way the reference is, as mutually
//                        recursive functions (move -> ability -> damage -> destroy -> ability ...), host only;
//   libproduct_host*.so  (-DORC_PRODUCT_CORE) the PRODUCT's rules core (monsoon_amd/csrc/rules.h: the explicit work
//                        stack the HIP kernels run) compiled for the host, so that its semantics can be checked against
//                        the golden vectors and against the oracle without a GPU.  Test infrastructure as well.
#if defined(ORC_PRODUCT_CORE)
#include "../monsoon_amd/csrc/rules.h"
#else
#include "recursive/rules.h"
#endif
#include "../monsoon_amd/csrc/canon.h"

using namespace msb;

namespace {

struct Game {
  alignas(16) uint8_t st[STATE_BYTES];
  uint32_t mt[MT_N];
  uint32_t out[2][MT_N];
  int cur;        // which of out[] is the current block
  uint32_t pos;   // cursor in the current block, 0..624
  int result;     // -2 running, -1 draw, 0 P1 win, 1 P2 win
  int steps;      // committed decisions
  uint64_t lookahead_steps;
  int la_fault;   // first capacity code (>= FAULT_CAPACITY) a look-ahead step of this game hit (the product's GameMeta.la_fault)
};

struct Oracle {
  std::vector<Game> games;
};

void refill(Game& g, int which) {
  mt_twist(g.mt);
  for (int i = 0; i < MT_N; i++) g.out[which][i] = mt_temper(g.mt[i]);
}

void seed_game(Game& g, uint32_t seed) {
  mt_seed(g.mt, seed);
  g.cur = 0;
  g.pos = 0;
  refill(g, 0);
  refill(g, 1);
}

// advance the game's stream to the cursor a step left in the record
void commit_rng(Game& g, Engine<FlatMem>& e) {
  g.pos = e.rng_pos();
  if (g.pos >= (uint32_t)MT_N) {
    g.pos -= MT_N;
    e.rng_block_advance();
    int old = g.cur;
    g.cur ^= 1;
    refill(g, old);
  }
}

// an engine over the game's record with the stream window attached at the game's cursor
Engine<FlatMem> engine(Game& g) {
  Engine<FlatMem> e;
  e.m.p = g.st;
  e.rng_attach(g.out[g.cur], g.out[g.cur ^ 1], g.pos);
  return e;
}

uint32_t peek(const Game& g) { return g.pos < (uint32_t)MT_N ? g.out[g.cur][g.pos] : g.out[g.cur ^ 1][g.pos - MT_N]; }

int game_result(Engine<FlatMem>& e) {
  // SURVEY §8c rollout contract (mirrors evo/fitness.py:160-166 scoring)
  int b0 = e.pl_base(0), b1 = e.pl_base(1);
  if (b1 < 0 && b0 >= 0) return 0;
  if (b0 < 0 && b1 >= 0) return 1;
  return -1;
}

// One decision of HeuristicAgent.select_action (evo/heuristic_agent.py:53-80): scores[a] for every
// legal action (ascending = the sorted legal list), argmax = first maximum.
int decide(Game& g, const double* w, double* scores_out, int* n_legal_out, uint64_t* mask_out) {
  Engine<FlatMem> e = engine(g);
  uint64_t mask[3];
  e.legal_mask(mask);
  if (mask_out) memcpy(mask_out, mask, sizeof(mask));
  double fb[10];
  bool before_raises = e.observation_raises();
  if (!before_raises) e.features(fb);
  int best = -1;
  double best_score = 0.0;
  int n_legal = 0;
  for (int a = 0; a < 156; a++) {
    if (!(mask[a >> 6] >> (a & 63) & 1)) {
      if (scores_out) scores_out[a] = NAN;
      continue;
    }
    n_legal++;
    Game c = g;  // copy.deepcopy(self.game), evo/game_adapter.py:284 (stream included)
    Engine<FlatMem> ce = engine(c);
    ce.step(a);
    g.lookahead_steps++;
    double s = 0.0;  // except Exception -> 0.0 (evo/heuristic_agent.py:48-51)
    if (ce.fault() >= FAULT_CAPACITY && !g.la_fault) g.la_fault = ce.fault();
    if (!ce.fault() && !before_raises && !ce.observation_raises()) {
      double fa[10];
      ce.features(fa);
      s = Engine<FlatMem>::action_score(w, fb, fa);
    }
    if (scores_out) scores_out[a] = s;
    if (best < 0 || s > best_score) {
      best = a;
      best_score = s;
    }
  }
  if (n_legal_out) *n_legal_out = n_legal;
  return best;
}

// Fault code of every legal action's look-ahead step (0 = none, 255 = not legal): lets a test tell a look-ahead the
// reference also rejects (FAULT_PY_EXCEPTION, score 0.0) from one this build flags as unsupported (FAULT_UNSUPPORTED).
void lookahead_faults(Game& g, uint8_t* out156) {
  Engine<FlatMem> e = engine(g);
  uint64_t mask[3];
  e.legal_mask(mask);
  for (int a = 0; a < 156; a++) {
    out156[a] = 255;
    if (!(mask[a >> 6] >> (a & 63) & 1)) continue;
    Game c = g;
    Engine<FlatMem> ce = engine(c);
    ce.step(a);
    int f = ce.fault();
    if (!f && ce.observation_raises()) f = FAULT_INT_CARD;
    out156[a] = (uint8_t)f;
  }
}

// adapter = adapter.apply_action(action): commit.  Returns the fault code of the step.
int commit(Game& g, int action) {
  Engine<FlatMem> e = engine(g);
  e.step(action);
  int f = e.fault();
  if (!f && e.observation_raises()) f = FAULT_INT_CARD;  // step() returns get_observation()
  commit_rng(g, e);
  g.steps++;
  return f;
}

}  // namespace

extern "C" {

void* orc_create(int n) {
  Oracle* o = new Oracle();
  o->games.resize(n);
  for (auto& g : o->games) {
    memset(&g, 0, sizeof(g));
    g.result = -2;
  }
  return o;
}
void orc_destroy(void* h) { delete (Oracle*)h; }
int orc_state_bytes() { return STATE_BYTES; }

int orc_reset(void* h, int gi, uint32_t seed, const uint8_t* deck0, const uint8_t* deck1, int f0, int f1) {
  Game& g = ((Oracle*)h)->games[gi];
  seed_game(g, seed);
  g.result = -2;
  g.steps = 0;
  g.lookahead_steps = 0;
  g.la_fault = 0;
  Engine<FlatMem> e = engine(g);
  e.init_game(deck0, deck1, f0, f1, seed);
  commit_rng(g, e);
  return e.fault();
}

void orc_legal(void* h, int gi, uint64_t* mask3) {
  Game& g = ((Oracle*)h)->games[gi];
  Engine<FlatMem> e = engine(g);
  e.legal_mask(mask3);
}

// Stormbound.step (games/stormbound.py:318-373).  Returns the fault code (0 = ok).
int orc_step(void* h, int gi, int action, int* reward, int* done) {
  Game& g = ((Oracle*)h)->games[gi];
  Engine<FlatMem> e = engine(g);
  int rd = e.step(action);
  if (reward) *reward = rd & 1;
  if (done) *done = (rd >> 1) & 1;
  int f = e.fault();
  commit_rng(g, e);
  g.steps++;
  return f;
}

// Stormbound.expert_action (consumes the game's stream).  Returns the action; *fault_out the fault code.
int orc_expert_action(void* h, int gi, int* fault_out) {
  Game& g = ((Oracle*)h)->games[gi];
  Engine<FlatMem> e = engine(g);
  int a = e.expert_action();
  if (fault_out) *fault_out = e.fault();
  commit_rng(g, e);
  return a;
}

int orc_observe(void* h, int gi, int32_t* out540) {
  Game& g = ((Oracle*)h)->games[gi];
  Engine<FlatMem> e = engine(g);
  if (e.observation_raises()) return 1;
  e.observe(out540);
  return 0;
}

int orc_features(void* h, int gi, double* f10) {
  Game& g = ((Oracle*)h)->games[gi];
  Engine<FlatMem> e = engine(g);
  if (e.observation_raises()) return 1;
  e.features(f10);
  return 0;
}

int orc_canon(void* h, int gi, uint8_t* buf) {
  Game& g = ((Oracle*)h)->games[gi];
  Engine<FlatMem> e = engine(g);
  return canon_record(e, peek(g), buf);
}
uint64_t orc_canon_hash(void* h, int gi) {
  uint8_t buf[CANON_MAX];
  int n = orc_canon(h, gi, buf);
  return fnv1a64(buf, n);
}

// fnv1a64 over the little-endian int32 observation bytes (0 where the observation raises)
uint64_t orc_obs_hash(void* h, int gi) {
  int32_t obs[540];
  if (orc_observe(h, gi, obs)) return 0;
  return fnv1a64((const uint8_t*)obs, sizeof(obs));
}

void orc_raw(void* h, int gi, uint8_t* buf) { memcpy(buf, ((Oracle*)h)->games[gi].st, STATE_BYTES); }

int orc_have_winner(void* h, int gi) {
  Game& g = ((Oracle*)h)->games[gi];
  Engine<FlatMem> e = engine(g);
  return e.have_winner() ? 1 : 0;
}
int orc_to_play(void* h, int gi) { return ((Oracle*)h)->games[gi].st[H_TOPLAY]; }
// out[0] = entity slots referenced by the last step (highest index + 1), out[1] = memory lists in use, out[2] = worlds in use
void orc_usage(void* h, int gi, int* out) {
  Game& g = ((Oracle*)h)->games[gi];
  Engine<FlatMem> e = engine(g);
  auto u = e.used_mask();
  int hi = 0;
  for (int i = 0; i < NUM_ENT; i++)
    if (u.has(i)) hi = i + 1;
  out[0] = hi;
  out[1] = out[2] = 0;
  for (int l = 0; l < REM_LISTS; l++) out[1] += g.st[OFF_REM + l * REM_LIST_BYTES + 1] ? 1 : 0;
  for (int w = 1; w <= WORLD_CAP; w++) out[2] += g.st[OFF_WORLD + (w - 1) * WORLD_BYTES + W_USED] ? 1 : 0;
}

// monsoon_game_faults of the product: the fault that stopped the game, else the first capacity code a look-ahead hit
int orc_game_fault(void* h, int gi) {
  Game& g = ((Oracle*)h)->games[gi];
  return g.st[H_FAULT] >= FAULT_CAPACITY ? g.st[H_FAULT] : (g.la_fault ? g.la_fault : g.st[H_FAULT]);
}

// scores156: NaN for illegal actions.  Returns the chosen action.
void orc_lookahead_faults(void* h, int gi, uint8_t* out156) { lookahead_faults(((Oracle*)h)->games[gi], out156); }

int orc_decide(void* h, int gi, const double* w10, double* scores156, uint64_t* mask3) {
  Game& g = ((Oracle*)h)->games[gi];
  int n;
  return decide(g, w10, scores156, &n, mask3);
}

// Full rollout per the SURVEY §8c contract.  Returns result (-1 draw, 0 P1, 1 P2); fills the
// per-decision trace if given (action u8, canon hash u64 after the commit).
int orc_rollout(void* h, int gi, const double* w_p1, const double* w_p2, int max_turns, uint8_t* trace_action,
                uint64_t* trace_hash, int* n_steps, uint64_t* n_lookahead, int* fault_out) {
  Game& g = ((Oracle*)h)->games[gi];
  int steps = 0, fault = 0;
  for (;;) {
    Engine<FlatMem> e = engine(g);
    if (e.have_winner() || steps >= max_turns) break;
    const double* w = g.st[H_TOPLAY] == 0 ? w_p1 : w_p2;
    int a = decide(g, w, nullptr, nullptr, nullptr);
    fault = commit(g, a);
    if (trace_action) trace_action[steps] = (uint8_t)a;
    if (trace_hash) trace_hash[steps] = orc_canon_hash(h, gi);
    steps++;
    if (fault) break;  // evo/fitness.py:208-210: exception during a turn -> break -> draw
  }
  Engine<FlatMem> e = engine(g);
  int result = -1;
  if (!fault && e.have_winner()) result = game_result(e);
  g.result = result;
  if (n_steps) *n_steps = steps;
  if (n_lookahead) *n_lookahead = g.lookahead_steps;
  if (fault_out) *fault_out = fault;
  return result;
}

// Multi-threaded batch of rollouts (bench.py cpu_baseline leg): games [0, n) of the handle must
// have been reset; the same weight vector plays both sides.  Returns total look-ahead steps.
uint64_t orc_rollout_batch(void* h, int n, const double* w, int max_turns, int n_threads, int8_t* results,
                           int32_t* steps, uint64_t* hashes) {
  std::atomic<int> next(0);
  std::atomic<uint64_t> total(0);
  auto worker = [&]() {
    for (;;) {
      int i = next.fetch_add(1);
      if (i >= n) break;
      int ns = 0, fl = 0;
      uint64_t nl = 0;
      int r = orc_rollout(h, i, w, w, max_turns, nullptr, nullptr, &ns, &nl, &fl);
      if (results) results[i] = (int8_t)r;
      if (steps) steps[i] = ns;
      if (hashes) hashes[i] = orc_canon_hash(h, i);
      total += nl;
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; t++) th.emplace_back(worker);
  for (auto& t : th) t.join();
  return total.load();
}

// A whole schedule of rollouts on n_threads host threads (the CPU side of the large GPU-vs-CPU comparisons): match k =
// {p1, p2, seed, deck} plays weights[p1] against weights[p2] on deck pair deck_pairs[deck]; per match: result, decisions,
// the fault orc_game_fault reports.  Returns the total look-ahead steps.
struct OrcMatch {
  int32_t p1, p2;
  uint32_t seed, deck;
};
uint64_t orc_rollout_schedule(const double* weights, const OrcMatch* matches, int n, const uint8_t* deck_pairs, int max_turns, int n_threads,
                              int8_t* results, int32_t* steps, uint8_t* faults) {
  std::atomic<int> next(0);
  std::atomic<uint64_t> total(0);
  auto worker = [&]() {
    void* h = orc_create(1);
    uint64_t mine = 0;
    for (;;) {
      int i = next.fetch_add(1);
      if (i >= n) break;
      const OrcMatch& m = matches[i];
      const uint8_t* d = deck_pairs + (size_t)m.deck * 24;
      int ns = 0, fl = 0;
      uint64_t nl = 0;
      int r = -1;
      if (orc_reset(h, 0, m.seed, d, d + 12, 0, 0) == 0) r = orc_rollout(h, 0, weights + (size_t)m.p1 * 10, weights + (size_t)m.p2 * 10, max_turns, nullptr, nullptr, &ns, &nl, &fl);
      results[i] = (int8_t)r;
      steps[i] = ns;
      faults[i] = (uint8_t)orc_game_fault(h, 0);
      mine += nl;
    }
    total += mine;
    orc_destroy(h);
  };
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; t++) th.emplace_back(worker);
  for (auto& t : th) t.join();
  return total.load();
}

// ---- scenario tests (tests/scenario_lib.py): the reference's own unit tests, recorded call by call ---------------
// Build game gi from a state stream, its numpy stream being RandomState(seed) advanced by stream_pos outputs.
int orc_scn_build(void* h, int gi, uint32_t seed, uint32_t stream_pos, const int32_t* state) {
  Game& g = ((Oracle*)h)->games[gi];
  seed_game(g, seed);
  g.result = -2;
  g.steps = 0;
  g.lookahead_steps = 0;
  g.la_fault = 0;
  memset(g.st, 0, sizeof(g.st));
  uint32_t blocks = stream_pos / MT_N;
  for (uint32_t b = 0; b < blocks; b++) {
    int old = g.cur;
    g.cur ^= 1;
    refill(g, old);
  }
  g.pos = stream_pos % MT_N;
  Engine<FlatMem> e = engine(g);
  if (REM_LISTS) e.m.st16(X_RNGBLK, (int)blocks);
  e.scn_build(state, seed);
  return e.fault();
}
// One recorded call.  log: {card, position} of every ability that ran, in order; returns the fault code.
int orc_scn_op(void* h, int gi, const int32_t* op, int32_t* log, int log_cap, int* n_log) {
  Game& g = ((Oracle*)h)->games[gi];
  Engine<FlatMem> e = engine(g);
  msb_trace_log = log;
  msb_trace_n = 0;
  msb_trace_cap = log ? log_cap : 0;
  int f = e.scn_op(op);
  if (n_log) *n_log = msb_trace_n;
  msb_trace_log = nullptr;
  msb_trace_cap = 0;
  commit_rng(g, e);
  return f;
}

// ---- RNG known-answer entry points (tests/golden/rng_kat.npz) ---------------------------------
void orc_rng_u32(uint32_t seed, int n, uint32_t* out) {
  Game g;
  memset(&g, 0, sizeof(g));
  seed_game(g, seed);
  for (int i = 0; i < n; i++) {
    Engine<FlatMem> e = engine(g);
    out[i] = e.rng_next_u32();
    commit_rng(g, e);
  }
}
void orc_rng_random(uint32_t seed, int n, double* out) {
  Game g;
  memset(&g, 0, sizeof(g));
  seed_game(g, seed);
  for (int i = 0; i < n; i++) {
    Engine<FlatMem> e = engine(g);
    out[i] = e.rng_random_sample();
    commit_rng(g, e);
  }
}
// out[i] = randint(0, bounds[i])
void orc_rng_randint(uint32_t seed, int n, const int* bounds, int* out) {
  Game g;
  memset(&g, 0, sizeof(g));
  seed_game(g, seed);
  for (int i = 0; i < n; i++) {
    Engine<FlatMem> e = engine(g);
    out[i] = e.rng_randint(0, bounds[i]);
    commit_rng(g, e);
  }
}
// shuffle(list(range(k))) repeated `reps` times on one stream
void orc_rng_shuffle(uint32_t seed, int k, int reps, int* out) {
  Game g;
  memset(&g, 0, sizeof(g));
  seed_game(g, seed);
  for (int r = 0; r < reps; r++) {
    Engine<FlatMem> e = engine(g);
    int* a = out + r * k;
    for (int i = 0; i < k; i++) a[i] = i;
    for (int i = k - 1; i >= 1; i--) {
      int j = (int)e.rng_interval((uint32_t)i);
      int t = a[i];
      a[i] = a[j];
      a[j] = t;
    }
    commit_rng(g, e);
  }
}
// monsoon_draw_decks on the CPU: RandomState(seed).choice(pool, 12, replace=False) twice per seed (numpy mtrand.pyx: choice ->
// permutation -> shuffle with random_interval draws).  Returns the number of seeds that ran past 1 248 outputs (the device's limit).
int orc_draw_decks(const uint32_t* seeds, int n, const uint8_t* pool, int pool_n, uint8_t* out) {
  int over = 0;
  for (int g = 0; g < n; g++) {
    uint32_t mt[MT_N], words[2 * MT_N];
    mt_seed(mt, seeds[g]);
    for (int b = 0; b < 2; b++) {
      mt_twist(mt);
      for (int k = 0; k < MT_N; k++) words[b * MT_N + k] = mt_temper(mt[k]);
    }
    int pos = 0;
    bool ov = false;
    for (int side = 0; side < 2; side++) {
      uint8_t perm[128];
      for (int i = 0; i < pool_n; i++) perm[i] = (uint8_t)i;
      for (int i = pool_n - 1; i >= 1; i--) {
        uint32_t mask = (uint32_t)i;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        uint32_t v = 0;
        do {
          if (pos >= 2 * MT_N) {
            ov = true;
            v = 0;
            break;
          }
          v = words[pos++] & mask;
        } while (v > (uint32_t)i);
        uint8_t t = perm[i];
        perm[i] = perm[v];
        perm[v] = t;
      }
      for (int k = 0; k < 12; k++) out[(size_t)g * 24 + side * 12 + k] = pool[perm[k]];
    }
    over += ov;
  }
  return over;
}
// monsoon_ga_offspring on the CPU (evo/population.py:75-89, evo/weights.py:12-40 over numpy's legacy global stream): the same
// walk of the stream as the device kernel, with the host's libm -- legacy_gauss calls libm's log and sqrt itself, so this
// equals numpy bit for bit except for exp (numpy's array exp is a SIMD routine of its own).
struct OrcNpState {
  uint32_t key[624];
  int32_t pos, has_gauss;
  double gauss;
};
static uint32_t np_u32(OrcNpState& s) {
  if (s.pos == MT_N) {
    mt_twist(s.key);
    s.pos = 0;
  }
  return mt_temper(s.key[s.pos++]);
}
static double np_double(OrcNpState& s) {
  uint32_t a = np_u32(s) >> 5, b = np_u32(s) >> 6;
  return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}
static double np_gauss(OrcNpState& s, long long& tries) {
  if (s.has_gauss) {
    const double t = s.gauss;
    s.has_gauss = 0;
    s.gauss = 0.0;
    return t;
  }
  double f, x1, x2, r2;
  do {
    x1 = 2.0 * np_double(s) - 1.0;
    x2 = 2.0 * np_double(s) - 1.0;
    r2 = x1 * x1 + x2 * x2;
    tries++;
  } while (r2 >= 1.0 || r2 == 0.0);
  f = sqrt(-2.0 * log(r2) / r2);
  s.gauss = f * x1;
  s.has_gauss = 1;
  return f * x2;
}
void orc_ga_offspring(OrcNpState* st, const double* pw, const double* ps, int mu, int dim, int lambda, double tau, double tau_prime,
                      double min_sigma, double* out_w, double* out_s, int32_t* out_parent, int64_t* out_tries) {
  OrcNpState& s = *st;
  long long tries = 0;
  for (int c = 0; c < lambda; c++) {
    uint32_t max = (uint32_t)(mu - 1), mask = max, v = 0;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    if (max) do { v = np_u32(s) & mask; } while (v > max);
    const int p = (int)v;
    for (int i = 0; i < dim; i++) (void)np_double(s);
    const double g = np_gauss(s, tries);
    double sig[16];
    for (int i = 0; i < dim; i++) {
      const double z = 0.0 + 1.0 * np_gauss(s, tries);
      const double a = tau_prime * g, b = tau * z;
      double sv = ps[(size_t)p * dim + i] * exp(a + b);
      sig[i] = sv > min_sigma ? sv : min_sigma;
      if (sv != sv) sig[i] = sv;
    }
    for (int i = 0; i < dim; i++) {
      const double stepv = 0.0 + sig[i] * np_gauss(s, tries);
      double w = pw[(size_t)p * dim + i] + stepv;
      w = w < 0.0 ? 0.0 : (w > 1.0 ? 1.0 : w);
      out_w[(size_t)c * dim + i] = w;
      out_s[(size_t)c * dim + i] = sig[i];
    }
    if (out_parent) out_parent[c] = p;
    if (out_tries) out_tries[c] = tries;
  }
}
// n values of RandomState(seed).normal(0, 1) and the number of stream outputs consumed after each (624 = a freshly seeded stream's position)
void orc_np_gauss(uint32_t seed, int n, double* values, int64_t* stream_pos) {
  OrcNpState s;
  mt_seed(s.key, seed);
  s.pos = MT_N;
  s.has_gauss = 0;
  s.gauss = 0.0;
  long long tries = 0, blocks = 0;
  for (int i = 0; i < n; i++) {
    const int before = s.pos;
    values[i] = 0.0 + 1.0 * np_gauss(s, tries);
    if (s.pos < before || (before == MT_N && s.pos != MT_N)) blocks++;
    stream_pos[i] = (blocks - 1) * MT_N + s.pos + MT_N;
  }
}
#if defined(MSB_COUNT_FRAMES)
long long* orc_frame_counts() { return msb_frame_count; }   // study build (scripts/frame_stats.py)
#endif
int orc_pyset_list(const uint8_t* keys, int n, uint8_t* out) { return pyset_list(keys, n, out); }
double orc_score(const double* w, const double* before, const double* after) {
  return Engine<FlatMem>::action_score(w, before, after);
}

}  // extern "C"
