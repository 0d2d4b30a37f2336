// Per-game state record: byte layout + the two memory accessors the rules core runs on.
//
// One record = STATE_BYTES bytes (a whole number of 16-byte granules), every field naturally aligned.  In HBM a
// batch of games is stored record-major: the hot kernel gives a wavefront to a game, so one game's record is one
// coalesced 16-bytes-per-lane read.  Inside a kernel a record lives in LDS; the candidate copies of a decision are
// interleaved across the lanes of the wave in 16-byte granules (granule c of lane l at (c*LANES + l)*16): lanes
// touching the same field hit different banks, and an entity (one granule) is a single ds_read_b128.
//
// What the fields restate (reference file:line):
//   header      games/stormbound.py:304 (player), board.py:20-25 (current_player, history, triggers,
//               is_resolving_trigger, phase)
//   board[20]   board.py:17  5x4 grid of object references -> entity slot ids (0xFF = None)
//   players[2]  player.py:13-37  indexed by PlayerOrder (FIRST=0, SECOND=1), not by local/remote
//   entities    unit.py:8-23 / structure.py:8-16 object attributes.  Slots have identity: the
//               reference iterates snapshots of entity OBJECTS (board.py:141-143) and keeps dead
//               units on its trigger stack, so a tile grid alone cannot reproduce it.
#pragma once
#include "msb_base.h"

namespace msb {

#if defined(MSB_CAP_ENT)
constexpr int NUM_ENT = MSB_CAP_ENT;   // capacity studies (scripts/c5_capacity.py)
#elif defined(MSB_EXT) && MSB_EXT == 2
constexpr int NUM_ENT = 254;  // the LARGE record: every slot id a byte can name (0xFE / 0xFF are markers)
#elif defined(MSB_EXT) && MSB_EXT
// 20 tiles + transient + b005's remembered copies + the entities of frozen world snapshots.  Round 3: 64 slots (2 832-byte record,
// 2 752 with 28 deck entries) instead of 128 (4 240): 8 instead of 4 candidate lanes fit a wavefront's LDS and the extended tier of a C5 generation runs 1.4x
// faster; 0.9 % instead of 0.2 % of random-deck games then need the large record (scripts/c5_capacity.py --ent 64), which the
// ladder gives them.
constexpr int NUM_ENT = 64;
#else
// 20 tiles + 4 transient (dead / displaced / spawned this step).  Round 3: 24 instead of 28 -- a 752-byte record, which
// together with the best successor parked in HBM lets 20 instead of 17 wavefronts share a CU's LDS.  No step of 68 000
// heuristic games on five deck families (N12M, S12, N12V, Ironclad vs Swarm, 20 000 random 107-card deck pairs) needs a
// 25th slot (22 slots: one game does, 21: 30 games; scripts/capacity_standard.py); a game that ever does reports
// FAULT_CAPACITY and the rollout path plays it again on the extended record (64 slots), like any other record limit.
constexpr int NUM_ENT = 24;
#endif
constexpr int HAND_CAP = 5;    // 4, transiently 5 (b305 returns itself to hand)
constexpr int DECK_SIZE = 12;  // cards per deck at construction (games/stormbound.py:295-302)
// Two builds of the same source.  The standard record holds every card except ua20 and b005; the
// EXTENDED record (-DMSB_EXT=1: libmonsoon_hip_ext.so / liboracle_ext.so) adds room for ua20's extra
// single-use deck cards (cards/ua20.py:27-32) and for b005's remembered deep copies (cards/b005.py:14-33).
#if defined(MSB_EXT) && MSB_EXT
#if MSB_EXT == 2
constexpr int DECK_CAP = 48;   // ua20 keeps appending single-use copies (cards/ua20.py:27-32): 2 of 131 072 C5 games pass 32 entries
#else
constexpr int DECK_CAP = 28;   // (32 until round 3: 64 bytes less per record let a seventh wavefront into a CU's LDS)
#endif
#if defined(MSB_CAP_REM)       // capacity studies (scripts/c5_capacity.py)
constexpr int REM_LISTS = MSB_CAP_REM, WORLD_CAP = MSB_CAP_WORLD;
#elif MSB_EXT == 2             // the LARGE record: where monsoon_rollout replays the games the extended record cannot hold
constexpr int REM_LISTS = 64, WORLD_CAP = 16;
#else
// 8 lists and 4 worlds (16 and 8 until late round 3): 2 400 instead of 2 752 bytes, an eighth wavefront in a CU's LDS; 1.03 %
// instead of 0.92 % of random-deck games then move on to the large record (scripts/c5_capacity.py --ent 64 --rem 8 --world 4)
constexpr int REM_LISTS = 8;   // memory lists (one per b005 with a pending memory, nested ones included)
constexpr int WORLD_CAP = 4;   // frozen world snapshots alive at once (see below)
#endif
static_assert(REM_LISTS <= 64 && WORLD_CAP < 32, "rem_collect keeps the live lists in a 64-bit and the live worlds in a 32-bit set");
#else
constexpr int DECK_CAP = 12;
constexpr int REM_LISTS = 0;
constexpr int WORLD_CAP = 0;
#endif
// b005 (cards/b005.py:14-33) remembers COPIES of its neighbours: a memory list holds up to 8 entity slots that are on
// no board.  Card.copy() is copy.deepcopy(self) with only the copy's own .player rebound (card.py:71-75), so
// everything else the copy reaches keeps pointing into a deep copy of the WHOLE GAME as it was at that moment: the
// entities inside a remembered b005's own memory belong to such a frozen "world" (their .player.board is the
// snapshot).  Restored later, they sit on the real board but move, fight and trigger on the snapshot until the next
// Board.flip re-binds every on-board entity to the real players (board.py:108-115).  A world = board + trigger stack
// + the scalar board/player fields an entity method can reach + its stream position; its entities are ordinary
// entity slots whose E_HOME names the world.  Entity methods run "in the world of self.player" (rules.h ctx_*).
#if defined(MSB_CAP_REMDEPTH)
constexpr int REM_DEPTH = MSB_CAP_REMDEPTH;
#elif defined(MSB_EXT) && MSB_EXT == 2
constexpr int REM_DEPTH = 12;
#else
constexpr int REM_DEPTH = 4;      // memories inside remembered copies inside ... : levels one deepcopy follows
#endif
constexpr int REM_PER_LIST = 8;   // surrounding tiles
constexpr int REM_LIST_BYTES = 4 + REM_PER_LIST;   // {n, used, pad2, REM_PER_LIST x entity slot}
constexpr int REM_NONE = 0xFF;
constexpr int REM_LOST = 0xFE;    // a remembered copy's own memory found no storage: using it raises FAULT_CAP_REM
// world storage (WORLD_BYTES each, worlds 1..WORLD_CAP; world 0 = the real game = the record's own fields)
constexpr int W_BOARD = 0;        // 20 x u8 slot
constexpr int W_TRIG = 20;        // TRIG_CAP x u8
constexpr int W_TOPLAY = 40, W_TRIG_N = 41, W_RESOLVING = 42, W_PHASE = 43, W_CP = 44, W_USED = 45, W_FRONT = 46;   // front: 2 x u8
constexpr int W_BASE = 48;        // 2 x i16
constexpr int W_RNG = 52;         // u32 absolute position in the game's stream (block * 624 + index)
constexpr int W_PARTIAL = 56;     // the snapshot is incomplete (entity slots ran out, or were taken back): entering it faults
constexpr int W_TRIGSRC = 60;     // u32, records with more than 128 slots only: the has_source bits of the trigger stack
constexpr int WORLD_BYTES = 64;
// Entity slots are a CACHE for the contents of frozen worlds: a world is only ever read if one of its entities is
// restored to the real board and acts before the next flip -- rare -- so a snapshot that does not fit is not an
// error by itself.  Its missing tiles hold SLOT_MISSING, a world that found no storage at all is WORLD_LOST, and
// only ENTERING such a world raises FAULT_CAPACITY.  When the real game needs a slot and none is free, the copies
// a world owns are taken back (that world becomes partial).
constexpr int SLOT_MISSING = 0xFE;
constexpr int WORLD_LOST = 0x7F;
constexpr int HOME_COPY = 0x80;   // E_HOME bit: the slot is a snapshot's private copy (it exists on that world's board only)
constexpr int TRIG_CAP = 20;
constexpr int PATH_CAP = 4;    // max movement among the cards (u069=4; gain_speed gives <= 3)
constexpr int SLOT_NONE = 0xFF;

// ---- header -----------------------------------------------------------------------------------
constexpr int H_TOPLAY = 0;     // 0 if Stormbound.player == 1 else 1; also the order of board.local
constexpr int H_FAULT = 1;
constexpr int H_HIST_N = 2;     // min(4, len(board.history))
constexpr int H_TRIG_N = 3;
constexpr int H_RESOLVING = 4;  // board.is_resolving_trigger
constexpr int H_PHASE = 5;
constexpr int H_DEPTH = 6;      // recursion guard (build limit)
constexpr int H_CP = 7;         // order of board.current_player
constexpr int H_HIST = 8;       // 4 x {owner, card}; oldest first
constexpr int H_USED = 16;      // u32: entity slots referenced since the start of this step
// The game's numpy stream as this record sees it (transient: set when the record is staged for a step, the
// cursor is written back by the caller): two resident blocks of tempered MT19937 outputs and a cursor.
constexpr int H_RNGPOS = 20;    // u16 index of the next word, 0..1247 (>= 624 reads the second block)
constexpr int H_OBSFAULT = 22;  // game flags: b0 a deck holds up01/up02/up03 (only then can get_observation raise, card.py:46);
                                //             b1 a card instance aliasing a board entity exists (b305 has returned to a hand)
constexpr int GF_OBSFAULT = 1, GF_ALIAS = 2;
constexpr int H_RNGOVER = 23;   // set when a step wanted more than the two resident blocks
constexpr int OFF_BOARD = 24;   // 20 x u8 slot
constexpr int OFF_TRIG = 44;    // TRIG_CAP x u8 (slot | has_source<<7; with more than 128 slots the flags move to X_TRIGSRC)
constexpr int H_RNGCUR = 64;    // u64 address of the current block of tempered outputs
constexpr int H_RNGNXT = 72;    // u64 address of the next block
constexpr int OFF_PL = 80;

// ---- player -------------------------------------------------------------------------------------
constexpr int P_BASE = 0;       // i16 Player.strength
constexpr int P_MANA = 2;       // i16 current_mana
constexpr int P_MAXMANA = 4;    // i16 max_mana
constexpr int P_FRONT = 6;      // u8 front_line
constexpr int P_FLAGS = 7;      // b0 replacable, b1 leftmost_movable
constexpr int P_FACTION = 8;
constexpr int P_HAND_N = 9;
constexpr int P_DECK_N = 10;
#if defined(MSB_EXT) && MSB_EXT
// Extended record: hand and deck are lists of OBJECT ids into a per-player card-instance table, because
// ua20's duplicate structures make list.remove() take a different (equal) object than the one drawn
// (player.py:52, structure.py:18-19): the same object can then sit in the hand and in the deck, or twice
// in the deck, and Player.reweight (player.py:57-59) touches it once per list position.
constexpr int INST_CAP = DECK_CAP + 8;   // >= HAND_CAP + DECK_CAP distinct objects can be listed
constexpr int P_HAND = 12;                       // HAND_CAP x u8 instance id
constexpr int P_DECK = P_HAND + HAND_CAP;        // DECK_CAP x u8 instance id
constexpr int P_INST = (P_DECK + DECK_CAP + 3) & ~3;          // INST_CAP x {card, cost, flags, x}
constexpr int P_AGE = P_INST + 4 * INST_CAP;                  // INST_CAP x u8 weight age (Card.weight = wtab[age], see below)
constexpr int P_IPOS = P_AGE + INST_CAP;                      // INST_CAP x u8: the .position a CF_STR object kept when it left the board
constexpr int IPOS_UNKNOWN = 0xFF;                            //   (Structure.__eq__ compares it, structure.py:18-19)
constexpr int PL_SIZE = (P_IPOS + INST_CAP + 15) & ~15;
#else
constexpr int P_HAND = 12;                      // HAND_CAP x {card, cost, flags, x}
constexpr int P_DECK = P_HAND + 4 * HAND_CAP;   // DECK_CAP x {card, cost, flags, x}
constexpr int P_AGE = P_DECK + 4 * DECK_CAP;    // DECK_CAP x u8 weight age of the card at that deck position
constexpr int PL_SIZE = (P_AGE + DECK_CAP + 15) & ~15;
#endif
static_assert((OFF_PL % 16) == 0 && (PL_SIZE % 16) == 0 && (P_AGE % 4) == 0, "granule / word alignment");
// Card.weight is never stored.  Every weight the reference can hold is f^k(1) with f(w) = w * 1.6 + 100 (player.py:31
// initial chain, :50 "choice.weight = 1", :57-59 reweight, cards/ua20.py:30 and cards/b305.py:41 "weight = 1"), so the
// record keeps the integer k ("age", one byte) and the f64 value comes from a table built with the same two rounded
// operations (msb_base.h g_wtab).  Player.reweight becomes a byte increment per deck entry.
constexpr int AGE_MAX = 255;
constexpr int WT_LDS_N = 32;     // table entries the kernels keep in LDS (older cards read the constant table)
#if defined(MSB_PROF) && MSB_PROF
constexpr int WT_LDS = 928;      // behind the function-scope counters of the profiling build (msb_base.h)
#else
constexpr int WT_LDS = 16;       // LDS address 0 is avoided (the null LDS pointer)
#endif
constexpr int WK_OVF_LDS = WT_LDS + 8 * WT_LDS_N;    // u64: global address of this workgroup's work-stack overflow block
constexpr int LDS_RECORDS = WK_OVF_LDS + 16;         // first LDS byte the kernels may use for records
// The work stack of the rules core (rules.h "Control flow"): up to SK_CAP 32-bit words per game.  On the device a lane's
// stack lives in SKW words of LDS (lane-interleaved like the record); whenever fewer than SK_NEED of them are free before
// a handler runs, everything below the top frame is EVICTED to the workgroup's block in HBM and comes back when the
// frames above it have run (rules.h wk_evict): a chain of more than a few nested abilities -- 2 % of the steps of a
// neutral-deck game.  A stack deeper than SK_CAP - SK_MARGIN words in all ends the step with FAULT_DEPTH (40 nested
// abilities / moves, where the reference's own recursion limit is restated, need about 530).
constexpr int SK_CAP = 640;
constexpr int SK_MARGIN = 32;
constexpr int SK_NEED = 12;   // the most one handler pushes before it returns to run() (an ability: F_AFTER 2 + the damage -> destroy -> next ability chain 6; F_EACH 8; F_TURN 7)
// card-instance flags (hand/deck entries {card, cost, flags, x}).  b305 puts the on-board structure OBJECT
// back into the hand (cards/b305.py:40-45): such an entry aliases entity slot x while that entity is
// on the board (CF_ALIAS) and keeps its last strength in x afterwards (CF_STR).  Both kinds have a
// position attribute, which matters for list.remove's equality (structure.py:18-19).
constexpr int CF_SINGLE_USE = 1, CF_FF = 2, CF_STR = 4, CF_ALIAS = 8;
// b4: x holds the card's printed strength (fresh card: cached so that features/observation need no card-table load);
// b5: the card is a Spell (observed strength -1, hand-quality strength 0)
constexpr int CF_XBASE = 16, CF_SPELL = 32;

// ---- entities: one 16-byte granule per slot (a whole entity is ONE ds_read_b128 / ds_write_b128) -----
constexpr int OFF_ENT = (OFF_PL + 2 * PL_SIZE + 15) & ~15;
constexpr int ENT_SIZE = 16;
constexpr int EO_CARD = 0;      // u8 card index (CARD_NONE = never used)
constexpr int EO_FLAGS = 1;     // u8 b0 owner order, b1 fixedly_forward, b2 resolving_play, b3 is_single_use
constexpr int EO_POS = 2;       // u8 recorded position y*4+x
constexpr int EO_MOV = 3;       // u8 movement
constexpr int EO_ST = 4;        // u8[5] status multiset counts (FROZEN, POISONED, CONFUSED, DISABLED, VITALIZED)
constexpr int EO_MOVEID = 9;    // u8 move_id (mod 256)
constexpr int EO_STR = 10;      // i16 strength
constexpr int EO_DMG = 12;      // i16 damage_taken
constexpr int EO_PATHN = 14;    // u8 len(path)
constexpr int EO_KIND = 15;     // u8 static card facts cached at creation: b0 is Unit, b1 overrides activate_ability, b2-5 trigger+1
constexpr int EK_UNIT = 1, EK_ABILITY = 2;
constexpr int E_PATH = OFF_ENT + ENT_SIZE * NUM_ENT;   // u32[NUM_ENT] packed path (PATH_CAP bytes)
constexpr int E_REM = E_PATH + 4 * NUM_ENT;                       // u8[NUM_ENT]: b005's list id (REM_NONE = [])
constexpr int E_HOME = E_REM + (REM_LISTS ? NUM_ENT : 0);          // u8[NUM_ENT]: the world the entity's .player belongs to
constexpr int OFF_REM = (E_HOME + (REM_LISTS ? NUM_ENT : 0) + 3) & ~3;   // REM_LISTS x REM_LIST_BYTES
constexpr int OFF_WORLD = (OFF_REM + REM_LISTS * REM_LIST_BYTES + 3) & ~3;   // WORLD_CAP x WORLD_BYTES
// extended-record scalars: X_CTX the world the engine is currently acting in; X_RNGBLK index of the record's current
// stream block (0 = first 624 outputs); X_SEED the game's seed (a world's old stream position may need a block that
// is no longer resident); X_USED_HI entity slots 32.. of H_USED (USED_WORDS - 1 more words); X_TRIGSRC the has_source
// bits of the trigger stack where a slot id needs all eight bits of its entry (TRIG_WIDE)
constexpr int USED_WORDS = (NUM_ENT + 31) / 32;
constexpr bool TRIG_WIDE = NUM_ENT > 128;
constexpr int TRIG_SLOT = TRIG_WIDE ? 0xff : 0x7f;   // the slot bits of a trigger-stack entry
constexpr int OFF_X = OFF_WORLD + WORLD_CAP * WORLD_BYTES;
constexpr int X_CTX = OFF_X, X_RNGBLK = OFF_X + 2, X_SEED = OFF_X + 4, X_USED_HI = OFF_X + 8;
constexpr int X_TRIGSRC = X_USED_HI + 4 * (USED_WORDS - 1);
constexpr int X_BYTES = REM_LISTS ? (X_TRIGSRC - OFF_X) + (TRIG_WIDE ? 4 : 0) : 0;
constexpr int STATE_BYTES = (OFF_X + X_BYTES + 15) & ~15;   // whole 16-byte granules
static_assert(NUM_ENT <= 254 && (!TRIG_WIDE || REM_LISTS), "slot ids are bytes; 0xFE and 0xFF are markers");
constexpr int STATE_WORDS = STATE_BYTES / 4;
constexpr int EF_OWNER = 1, EF_FF = 2, EF_RESOLVING_PLAY = 4, EF_SINGLE_USE = 8;
typedef uint32_t msb_u32x4 __attribute__((vector_size(16)));

// ---- accessors ----------------------------------------------------------------------------------
#if !defined(__HIPCC__)
static thread_local int32_t* msb_trace_log = nullptr;   // {card, position} pairs (FlatMem::trace_ability)
static thread_local int msb_trace_n = 0, msb_trace_cap = 0;
static thread_local uint32_t msb_host_wk[640];   // SK_CAP words: the work stack of the game this thread is stepping
static thread_local uint32_t msb_host_ovf[640];  // its eviction block (only the MSB_HOST_SKW build ever evicts)
#endif
// Host / flat: the record is a contiguous byte array.
struct FlatMem {
  uint8_t* p;
  MSB_HD MSB_INL int ld8(int o) const { return p[o]; }
  MSB_HD MSB_INL void st8(int o, int v) { p[o] = (uint8_t)v; }
  MSB_HD MSB_INL int ld16(int o) const { return *(const int16_t*)(p + o); }
  MSB_HD MSB_INL void st16(int o, int v) { *(int16_t*)(p + o) = (int16_t)v; }
  MSB_HD MSB_INL uint32_t ld32(int o) const { return *(const uint32_t*)(p + o); }
  MSB_HD MSB_INL void st32(int o, uint32_t v) { *(uint32_t*)(p + o) = v; }
  MSB_HD MSB_INL double ldf(int o) const { return *(const double*)(p + o); }
  MSB_HD MSB_INL void stf(int o, double v) { *(double*)(p + o) = v; }
  MSB_HD MSB_INL uint64_t ld64(int o) const { return *(const uint64_t*)(p + o); }
  MSB_HD MSB_INL void st64(int o, uint64_t v) { *(uint64_t*)(p + o) = v; }
  // the host record is only 8-byte aligned: no aligned vector moves
  MSB_HD MSB_INL msb_u32x4 ld128(int o) const {
    msb_u32x4 v;
    __builtin_memcpy(&v, p + o, 16);
    return v;
  }
  MSB_HD MSB_INL void st128(int o, msb_u32x4 v) { __builtin_memcpy(p + o, &v, 16); }
  // (granule, byte within the granule) forms
  MSB_HD MSB_INL int ld8g(int g, int k) const { return ld8(g * 16 + k); }
  MSB_HD MSB_INL void st8g(int g, int k, int v) { st8(g * 16 + k, v); }
  MSB_HD MSB_INL int ld16g(int g, int k) const { return ld16(g * 16 + k); }
  MSB_HD MSB_INL void st16g(int g, int k, int v) { st16(g * 16 + k, v); }
  MSB_HD MSB_INL msb_u32x4 ld128g(int g) const { return ld128(g * 16); }
  MSB_HD MSB_INL void st128g(int g, msb_u32x4 v) { st128(g * 16, v); }
  MSB_HD MSB_INL uint32_t ld32g(int g, int k) const { return ld32(g * 16 + k); }
  MSB_HD MSB_INL void st32g(int g, int k, uint32_t v) { st32(g * 16 + k, v); }
  MSB_HD MSB_INL double ldfg(int g, int k) const { return ldf(g * 16 + k); }
  MSB_HD MSB_INL void stfg(int g, int k, double v) { stf(g * 16 + k, v); }
  MSB_HD MSB_INL static double wtab(int age) { return g_wtab.v[age & AGE_MAX]; }
  // work stack: a per-thread array on the host, large enough never to evict; the device never steps a game through
  // this accessor
#if defined(MSB_HOST_SKW)   // study / test build: the host evicts like the device does (oracle/Makefile libproduct_host_evict.so)
  static constexpr int SKW = MSB_HOST_SKW;
#else
  static constexpr int SKW = SK_CAP;
#endif
#if !defined(__HIPCC__)
  MSB_HD MSB_INL static uint32_t sk_ld(int i) { return msb_host_wk[i]; }
  MSB_HD MSB_INL static void sk_st(int i, uint32_t v) {
#if defined(MSB_HOST_SKW)
    if (i >= SKW) __builtin_trap();   // a handler pushed more than SK_NEED words: the device would write past its LDS stack
#endif
    msb_host_wk[i] = v;
  }
  MSB_HD MSB_INL static uint32_t ovf_ld(int i) { return msb_host_ovf[i]; }
  MSB_HD MSB_INL static void ovf_st(int i, uint32_t v) { msb_host_ovf[i] = v; }
#else
  MSB_HD MSB_INL static uint32_t sk_ld(int) { return 0; }
  MSB_HD MSB_INL static void sk_st(int, uint32_t) {}
  MSB_HD MSB_INL static uint32_t ovf_ld(int) { return 0; }
  MSB_HD MSB_INL static void ovf_st(int, uint32_t) {}
#endif
  // order in which abilities run (scenario tests): a log the host oracle can switch on; compiled out of device code
  MSB_HD MSB_INL static void trace_ability(int card, int pos) {
#if !defined(__HIPCC__)
    if (msb_trace_log && msb_trace_n < msb_trace_cap) {
      msb_trace_log[2 * msb_trace_n] = card;
      msb_trace_log[2 * msb_trace_n + 1] = pos;
      msb_trace_n++;
    }
#else
    (void)card;
    (void)pos;
#endif
  }
};

#if defined(__HIPCC__)
// Device-only forms with explicit address spaces, so that hipcc emits ds_read/ds_write (LDS) and
// global_load/global_store (HBM) instead of flat_* instructions behind non-inlined calls.
#define MSB_AS_LDS __attribute__((address_space(3)))
#define MSB_AS_GLB __attribute__((address_space(1)))
// weight table lookup of the LDS accessors: the first WT_LDS_N entries sit at LDS address WT_LDS (every kernel that
// runs the rules core fills them, lds_init_wtab), older cards read the constant table
MSB_HD MSB_INL double lds_wtab(int age) {
  if (age < WT_LDS_N) return *(MSB_AS_LDS const double*)(uintptr_t)(WT_LDS + 8 * age);
  return g_wtab.v[age & AGE_MAX];
}
// call with all threads of the workgroup, before any engine code.  wk_ovf: this workgroup's overflow block of the work
// stack (LANES * SK_CAP words, see LaneMem::ovf_*), or null where no game is stepped
MSB_HD MSB_INL void lds_init_wtab(uint32_t* wk_ovf = nullptr) {
  for (int i = (int)__builtin_amdgcn_workitem_id_x(); i < WT_LDS_N; i += (int)__builtin_amdgcn_workgroup_size_x())
    *(MSB_AS_LDS double*)(uintptr_t)(WT_LDS + 8 * i) = g_wtab.v[i];
  if (__builtin_amdgcn_workitem_id_x() == 0) *(MSB_AS_LDS uint64_t*)(uintptr_t)WK_OVF_LDS = (uint64_t)(uintptr_t)wk_ovf;
  __syncthreads();
}
// LDS image of one record, interleaved across the lanes of a wave in 16-BYTE granules: granule c of
// lane l sits at (c*LANES + l)*16.  A lane's record is copied 16 bytes at a time (ds_read_b128 /
// ds_write_b128, conflict-free: consecutive lanes touch consecutive 16-byte slots), f64 weights are one
// ds_read_b64, and with 8 lanes per game a same-field access by all lanes still hits 8 distinct banks.
//
// The accessors are STATELESS: the kernels use only dynamic LDS, so the image starts at LDS address
// BASE (a compile-time constant) and the lane is the work-item id.  An Engine over such an accessor is an
// empty object -- nothing has to be reloaded through `this` behind the non-inlined (recursive) calls of
// the rules core, which cost a flat_load round trip per call when the accessor held a pointer.
//
// SKB / SKW_: LDS address and words per lane of the work stack, interleaved across the lanes word by word (word i of
// lane l at SKB + (i*LANES + l)*4).  ovf_*: word i of the lane's share of the workgroup's eviction block in HBM
// (word-major as well: lanes at the same depth touch neighbouring words).
template <int LANES, int BASE, int SKB = 0, int SKW_ = 0>
struct LaneMem {   // this lane's private record among LANES interleaved ones
  static constexpr int SKW = SKW_;
  MSB_HD MSB_INL static uint32_t* ovf(int i) {
    uint32_t* base = (uint32_t*)(uintptr_t)(*(MSB_AS_LDS const uint64_t*)(uintptr_t)WK_OVF_LDS);
    return base + (size_t)i * LANES + (int)__builtin_amdgcn_workitem_id_x();
  }
  MSB_HD MSB_INL static uint32_t ovf_ld(int i) { return *ovf(i); }
  MSB_HD MSB_INL static void ovf_st(int i, uint32_t v) { *ovf(i) = v; }
  MSB_HD MSB_INL static uint32_t sk_ld(int i) { return *(MSB_AS_LDS const uint32_t*)(uintptr_t)(SKB + (i * LANES + (int)__builtin_amdgcn_workitem_id_x()) * 4); }
  MSB_HD MSB_INL static void sk_st(int i, uint32_t v) { *(MSB_AS_LDS uint32_t*)(uintptr_t)(SKB + (i * LANES + (int)__builtin_amdgcn_workitem_id_x()) * 4) = v; }
  MSB_HD MSB_INL static MSB_AS_LDS uint8_t* b(int o) {
    return (MSB_AS_LDS uint8_t*)(uintptr_t)(BASE + (o >> 4) * (LANES * 16) + (o & 15) + (int)__builtin_amdgcn_workitem_id_x() * 16);
  }
  MSB_HD MSB_INL static int ld8(int o) { return *b(o); }
  MSB_HD MSB_INL static void st8(int o, int v) { *b(o) = (uint8_t)v; }
  MSB_HD MSB_INL static int ld16(int o) { return *(MSB_AS_LDS const int16_t*)b(o); }
  MSB_HD MSB_INL static void st16(int o, int v) { *(MSB_AS_LDS int16_t*)b(o) = (int16_t)v; }
  MSB_HD MSB_INL static uint32_t ld32(int o) { return *(MSB_AS_LDS const uint32_t*)b(o); }
  MSB_HD MSB_INL static void st32(int o, uint32_t v) { *(MSB_AS_LDS uint32_t*)b(o) = v; }
  MSB_HD MSB_INL static double ldf(int o) { return *(MSB_AS_LDS const double*)b(o); }
  MSB_HD MSB_INL static void stf(int o, double v) { *(MSB_AS_LDS double*)b(o) = v; }
  MSB_HD MSB_INL static uint64_t ld64(int o) { return *(MSB_AS_LDS const uint64_t*)b(o); }
  MSB_HD MSB_INL static void st64(int o, uint64_t v) { *(MSB_AS_LDS uint64_t*)b(o) = v; }
  MSB_HD MSB_INL static msb_u32x4 ld128(int o) { return *(MSB_AS_LDS const msb_u32x4*)b(o); }
  MSB_HD MSB_INL static void st128(int o, msb_u32x4 v) { *(MSB_AS_LDS msb_u32x4*)b(o) = v; }
  // (granule, byte within the granule): granule * (LANES*16) + lane*16 + BASE + k -- one shift-add for a dynamic
  // granule, the rest folds into the instruction's immediate offset
  MSB_HD MSB_INL static MSB_AS_LDS uint8_t* gb(int g, int k) {
    return (MSB_AS_LDS uint8_t*)(uintptr_t)(BASE + g * (LANES * 16) + k + (int)__builtin_amdgcn_workitem_id_x() * 16);
  }
  MSB_HD MSB_INL static int ld8g(int g, int k) { return *gb(g, k); }
  MSB_HD MSB_INL static void st8g(int g, int k, int v) { *gb(g, k) = (uint8_t)v; }
  MSB_HD MSB_INL static int ld16g(int g, int k) { return *(MSB_AS_LDS const int16_t*)gb(g, k); }
  MSB_HD MSB_INL static void st16g(int g, int k, int v) { *(MSB_AS_LDS int16_t*)gb(g, k) = (int16_t)v; }
  MSB_HD MSB_INL static msb_u32x4 ld128g(int g) { return *(MSB_AS_LDS const msb_u32x4*)gb(g, 0); }
  MSB_HD MSB_INL static void st128g(int g, msb_u32x4 v) { *(MSB_AS_LDS msb_u32x4*)gb(g, 0) = v; }
  MSB_HD MSB_INL static uint32_t ld32g(int g, int k) { return *(MSB_AS_LDS const uint32_t*)gb(g, k); }
  MSB_HD MSB_INL static void st32g(int g, int k, uint32_t v) { *(MSB_AS_LDS uint32_t*)gb(g, k) = v; }
  MSB_HD MSB_INL static double ldfg(int g, int k) { return *(MSB_AS_LDS const double*)gb(g, k); }
  MSB_HD MSB_INL static void stfg(int g, int k, double v) { *(MSB_AS_LDS double*)gb(g, k) = v; }
  MSB_HD MSB_INL static double wtab(int age) { return lds_wtab(age); }
  MSB_HD MSB_INL static void trace_ability(int, int) {}
};
// The same with the ability log of the scenario tests switched on: {card, position} pairs appended to an LDS array
// (count at TRACE_BASE, pairs behind it).  Only the one-lane diagnostics kernel of monsoon_debug_op uses it.
template <int LANES, int BASE, int SKB, int SKW, int TRACE_BASE, int TRACE_CAP>
struct TraceLaneMem : LaneMem<LANES, BASE, SKB, SKW> {
  MSB_HD MSB_INL static void trace_ability(int card, int pos) {
    MSB_AS_LDS int32_t* t = (MSB_AS_LDS int32_t*)(uintptr_t)TRACE_BASE;
    int n = t[0];
    if (n < TRACE_CAP) {
      t[1 + 2 * n] = card;
      t[2 + 2 * n] = pos;
      t[0] = n + 1;
    }
  }
};
template <int BASE>
struct SharedMem {   // one contiguous record read by every lane of the wave (LDS broadcast)
  MSB_HD MSB_INL static MSB_AS_LDS uint8_t* b(int o) { return (MSB_AS_LDS uint8_t*)(uintptr_t)(BASE + o); }
  MSB_HD MSB_INL static int ld8(int o) { return *b(o); }
  MSB_HD MSB_INL static void st8(int o, int v) { *b(o) = (uint8_t)v; }
  MSB_HD MSB_INL static int ld16(int o) { return *(MSB_AS_LDS const int16_t*)b(o); }
  MSB_HD MSB_INL static void st16(int o, int v) { *(MSB_AS_LDS int16_t*)b(o) = (int16_t)v; }
  MSB_HD MSB_INL static uint32_t ld32(int o) { return *(MSB_AS_LDS const uint32_t*)b(o); }
  MSB_HD MSB_INL static void st32(int o, uint32_t v) { *(MSB_AS_LDS uint32_t*)b(o) = v; }
  MSB_HD MSB_INL static double ldf(int o) { return *(MSB_AS_LDS const double*)b(o); }
  MSB_HD MSB_INL static void stf(int o, double v) { *(MSB_AS_LDS double*)b(o) = v; }
  MSB_HD MSB_INL static uint64_t ld64(int o) { return *(MSB_AS_LDS const uint64_t*)b(o); }
  MSB_HD MSB_INL static void st64(int o, uint64_t v) { *(MSB_AS_LDS uint64_t*)b(o) = v; }
  MSB_HD MSB_INL static msb_u32x4 ld128(int o) { return *(MSB_AS_LDS const msb_u32x4*)b(o); }
  MSB_HD MSB_INL static void st128(int o, msb_u32x4 v) { *(MSB_AS_LDS msb_u32x4*)b(o) = v; }
  MSB_HD MSB_INL static int ld8g(int g, int k) { return ld8(g * 16 + k); }
  MSB_HD MSB_INL static void st8g(int g, int k, int v) { st8(g * 16 + k, v); }
  MSB_HD MSB_INL static int ld16g(int g, int k) { return ld16(g * 16 + k); }
  MSB_HD MSB_INL static void st16g(int g, int k, int v) { st16(g * 16 + k, v); }
  MSB_HD MSB_INL static msb_u32x4 ld128g(int g) { return ld128(g * 16); }
  MSB_HD MSB_INL static void st128g(int g, msb_u32x4 v) { st128(g * 16, v); }
  MSB_HD MSB_INL static uint32_t ld32g(int g, int k) { return ld32(g * 16 + k); }
  MSB_HD MSB_INL static void st32g(int g, int k, uint32_t v) { st32(g * 16 + k, v); }
  MSB_HD MSB_INL static double ldfg(int g, int k) { return ldf(g * 16 + k); }
  MSB_HD MSB_INL static void stfg(int g, int k, double v) { stf(g * 16 + k, v); }
  MSB_HD MSB_INL static double wtab(int age) { return lds_wtab(age); }
  MSB_HD MSB_INL static void trace_ability(int, int) {}
  // the shared copy is only read (legal mask, features): nothing is ever stepped through it
  static constexpr int SKW = SK_CAP;
  MSB_HD MSB_INL static uint32_t sk_ld(int) { return 0; }
  MSB_HD MSB_INL static void sk_st(int, uint32_t) {}
  MSB_HD MSB_INL static uint32_t ovf_ld(int) { return 0; }
  MSB_HD MSB_INL static void ovf_st(int, uint32_t) {}
};
// Column 0 of a lane-interleaved image (LaneMem<LANES, BASE, ..>), addressed by every lane alike: where kernels_reg.h keeps
// the LDS image of the game's current record (the record itself lives in registers there).  Read-mostly like SharedMem.
template <int LANES, int BASE>
struct Col0Mem {
  MSB_HD MSB_INL static MSB_AS_LDS uint8_t* b(int o) { return (MSB_AS_LDS uint8_t*)(uintptr_t)(BASE + (o >> 4) * (LANES * 16) + (o & 15)); }
  MSB_HD MSB_INL static MSB_AS_LDS uint8_t* gb(int g, int k) { return (MSB_AS_LDS uint8_t*)(uintptr_t)(BASE + g * (LANES * 16) + k); }
  MSB_HD MSB_INL static int ld8(int o) { return *b(o); }
  MSB_HD MSB_INL static void st8(int o, int v) { *b(o) = (uint8_t)v; }
  MSB_HD MSB_INL static int ld16(int o) { return *(MSB_AS_LDS const int16_t*)b(o); }
  MSB_HD MSB_INL static void st16(int o, int v) { *(MSB_AS_LDS int16_t*)b(o) = (int16_t)v; }
  MSB_HD MSB_INL static uint32_t ld32(int o) { return *(MSB_AS_LDS const uint32_t*)b(o); }
  MSB_HD MSB_INL static void st32(int o, uint32_t v) { *(MSB_AS_LDS uint32_t*)b(o) = v; }
  MSB_HD MSB_INL static double ldf(int o) { return *(MSB_AS_LDS const double*)b(o); }
  MSB_HD MSB_INL static void stf(int o, double v) { *(MSB_AS_LDS double*)b(o) = v; }
  MSB_HD MSB_INL static uint64_t ld64(int o) { return *(MSB_AS_LDS const uint64_t*)b(o); }
  MSB_HD MSB_INL static void st64(int o, uint64_t v) { *(MSB_AS_LDS uint64_t*)b(o) = v; }
  MSB_HD MSB_INL static msb_u32x4 ld128(int o) { return *(MSB_AS_LDS const msb_u32x4*)b(o); }
  MSB_HD MSB_INL static void st128(int o, msb_u32x4 v) { *(MSB_AS_LDS msb_u32x4*)b(o) = v; }
  MSB_HD MSB_INL static int ld8g(int g, int k) { return *gb(g, k); }
  MSB_HD MSB_INL static void st8g(int g, int k, int v) { *gb(g, k) = (uint8_t)v; }
  MSB_HD MSB_INL static int ld16g(int g, int k) { return *(MSB_AS_LDS const int16_t*)gb(g, k); }
  MSB_HD MSB_INL static void st16g(int g, int k, int v) { *(MSB_AS_LDS int16_t*)gb(g, k) = (int16_t)v; }
  MSB_HD MSB_INL static msb_u32x4 ld128g(int g) { return *(MSB_AS_LDS const msb_u32x4*)gb(g, 0); }
  MSB_HD MSB_INL static void st128g(int g, msb_u32x4 v) { *(MSB_AS_LDS msb_u32x4*)gb(g, 0) = v; }
  MSB_HD MSB_INL static uint32_t ld32g(int g, int k) { return *(MSB_AS_LDS const uint32_t*)gb(g, k); }
  MSB_HD MSB_INL static void st32g(int g, int k, uint32_t v) { *(MSB_AS_LDS uint32_t*)gb(g, k) = v; }
  MSB_HD MSB_INL static double ldfg(int g, int k) { return *(MSB_AS_LDS const double*)gb(g, k); }
  MSB_HD MSB_INL static void stfg(int g, int k, double v) { *(MSB_AS_LDS double*)gb(g, k) = v; }
  MSB_HD MSB_INL static double wtab(int age) { return lds_wtab(age); }
  MSB_HD MSB_INL static void trace_ability(int, int) {}
  static constexpr int SKW = SK_CAP;   // (nothing is stepped through it)
  MSB_HD MSB_INL static uint32_t sk_ld(int) { return 0; }
  MSB_HD MSB_INL static void sk_st(int, uint32_t) {}
  MSB_HD MSB_INL static uint32_t ovf_ld(int) { return 0; }
  MSB_HD MSB_INL static void ovf_st(int, uint32_t) {}
};
// The same for a wavefront that plays several games at once (kernels_multi.h): lanes [k*U, (k+1)*U) belong to game slot k,
// whose current record sits at BASE + k*STRIDE.
template <int BASE, int STRIDE, int U>
struct GroupMem {
  MSB_HD MSB_INL static MSB_AS_LDS uint8_t* b(int o) {
    return (MSB_AS_LDS uint8_t*)(uintptr_t)(BASE + ((int)__builtin_amdgcn_workitem_id_x() / U) * STRIDE + o);
  }
  MSB_HD MSB_INL static int ld8(int o) { return *b(o); }
  MSB_HD MSB_INL static void st8(int o, int v) { *b(o) = (uint8_t)v; }
  MSB_HD MSB_INL static int ld16(int o) { return *(MSB_AS_LDS const int16_t*)b(o); }
  MSB_HD MSB_INL static void st16(int o, int v) { *(MSB_AS_LDS int16_t*)b(o) = (int16_t)v; }
  MSB_HD MSB_INL static uint32_t ld32(int o) { return *(MSB_AS_LDS const uint32_t*)b(o); }
  MSB_HD MSB_INL static void st32(int o, uint32_t v) { *(MSB_AS_LDS uint32_t*)b(o) = v; }
  MSB_HD MSB_INL static double ldf(int o) { return *(MSB_AS_LDS const double*)b(o); }
  MSB_HD MSB_INL static void stf(int o, double v) { *(MSB_AS_LDS double*)b(o) = v; }
  MSB_HD MSB_INL static uint64_t ld64(int o) { return *(MSB_AS_LDS const uint64_t*)b(o); }
  MSB_HD MSB_INL static void st64(int o, uint64_t v) { *(MSB_AS_LDS uint64_t*)b(o) = v; }
  MSB_HD MSB_INL static msb_u32x4 ld128(int o) { return *(MSB_AS_LDS const msb_u32x4*)b(o); }
  MSB_HD MSB_INL static void st128(int o, msb_u32x4 v) { *(MSB_AS_LDS msb_u32x4*)b(o) = v; }
  MSB_HD MSB_INL static int ld8g(int g, int k) { return ld8(g * 16 + k); }
  MSB_HD MSB_INL static void st8g(int g, int k, int v) { st8(g * 16 + k, v); }
  MSB_HD MSB_INL static int ld16g(int g, int k) { return ld16(g * 16 + k); }
  MSB_HD MSB_INL static void st16g(int g, int k, int v) { st16(g * 16 + k, v); }
  MSB_HD MSB_INL static msb_u32x4 ld128g(int g) { return ld128(g * 16); }
  MSB_HD MSB_INL static void st128g(int g, msb_u32x4 v) { st128(g * 16, v); }
  MSB_HD MSB_INL static uint32_t ld32g(int g, int k) { return ld32(g * 16 + k); }
  MSB_HD MSB_INL static void st32g(int g, int k, uint32_t v) { st32(g * 16 + k, v); }
  MSB_HD MSB_INL static double ldfg(int g, int k) { return ldf(g * 16 + k); }
  MSB_HD MSB_INL static void stfg(int g, int k, double v) { stf(g * 16 + k, v); }
  MSB_HD MSB_INL static double wtab(int age) { return lds_wtab(age); }
  MSB_HD MSB_INL static void trace_ability(int, int) {}
  static constexpr int SKW = SK_CAP;   // (read only, like SharedMem: nothing is stepped through it)
  MSB_HD MSB_INL static uint32_t sk_ld(int) { return 0; }
  MSB_HD MSB_INL static void sk_st(int, uint32_t) {}
  MSB_HD MSB_INL static uint32_t ovf_ld(int) { return 0; }
  MSB_HD MSB_INL static void ovf_st(int, uint32_t) {}
};
#endif

template <class A, class B>
MSB_HD MSB_INL void copy_state(A& dst, const B& src) {
  for (int w = 0; w < STATE_WORDS; w++) dst.st32(w * 4, src.ld32(w * 4));
}

}  // namespace msb
