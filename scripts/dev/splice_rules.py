"""One-off: builds monsoon_amd/csrc/rules.h (explicit work stack) from oracle/recursive/rules.h (the recursive core)
by replacing the functions that sit on a call cycle.  Kept for the record of what was replaced; not part of any build."""
import re
import sys

SRC = "oracle/recursive/rules.h"
DST = "monsoon_amd/csrc/rules.h"
lines = open(SRC).read().split("\n")

B = {}

B[(10, 12)] = r'''// Control flow: the reference recurses (move -> ability -> deal_damage -> destroy -> ability -> command -> move ...).
// Here every function on such a cycle is a FRAME on an explicit per-game work stack and Engine::run() is the only
// loop (see "Control flow" below): no call cycle is left in the C++, so the device code needs no dynamic stack.'''

B[(956, 1005)] = r'''  // copy.deepcopy of a memory list (the list object, its entities, their own memories, and -- through entity.player
  // -- the world each of them belongs to, once per deepcopy call: memo[]).  The reference's recursion over nested
  // memories is an explicit depth-first walk here (pre-order, so lists, slots and worlds are handed out in the same
  // order): level L copies list s_src[L] into s_dst[L], s_k[L] is its next element and s_c[L] the copy whose own memory
  // the level below is making.
  MSB_HD MSB_NOINLINE int rem_deep_copy(int src0, int* memo, int root_old, int root_new) {
    int s_src[REM_DEPTH + 2], s_dst[REM_DEPTH + 2], s_k[REM_DEPTH + 2], s_c[REM_DEPTH + 2];
    int L = 0, ret = REM_NONE;
    enum { CALL, LOOP, RET } mode = CALL;
    s_src[0] = src0;
    for (;;) {
      if (mode == CALL) {   // deepcopy of list s_src[L] begins
        const int src = s_src[L];
        if (L > REM_DEPTH) {
          set_fault(FAULT_CAP_REM);
          ret = REM_NONE;
          mode = RET;
          continue;
        }
        if (src == REM_LOST) {
          ret = REM_LOST;
          mode = RET;
          continue;
        }
        const int dst = rem_alloc_soft();
        if (dst == REM_LOST) {   // no list storage: the copy's memory is lost unless it is never used
          ret = REM_LOST;
          mode = RET;
          continue;
        }
        m.st8(rem_off(dst), rem_n(src));
        s_dst[L] = dst;
        s_k[L] = 0;
        mode = LOOP;
      } else if (mode == LOOP) {   // next element of the list being copied at level L
        const int src = s_src[L], dst = s_dst[L], k = s_k[L];
        if (k >= rem_n(src) || fault()) {
          ret = dst;
          mode = RET;
          continue;
        }
        const int r = rem_get(src, k);
        const int c = dup_entity(r);
        if (fault()) {
          ret = dst;
          mode = RET;
          continue;
        }
        m.st8(rem_off(dst) + 4 + k, c);
        const int H = m.ld8(E_HOME + r) & 0x7f;
        int nh = WORLD_LOST;
        if (H != WORLD_LOST) {
          if (memo[H] < 0) memo[H] = world_snapshot(H, root_old, root_new);
          if (fault()) {
            ret = dst;
            mode = RET;
            continue;
          }
          nh = memo[H];
        }
        m.st8(E_HOME + c, nh);
        s_k[L] = k + 1;
        const int L2 = m.ld8(E_REM + r);
        if (L2 == REM_LOST || (L2 != REM_NONE && rem_n(L2) > 0)) {
          s_c[L] = c;
          L++;
          s_src[L] = L2;
          mode = CALL;
        }
      } else {   // level L is done: `ret` is its copy
        if (L == 0) return ret;
        L--;
        if (fault()) {   // the level above breaks out of its loop and hands back what it has
          ret = s_dst[L];
          continue;
        }
        m.st8(E_REM + s_c[L], ret);
        mode = LOOP;
      }
    }
  }
  // Card.copy() of the on-board entity e (card.py:71-75): deepcopy, then copied.player = self.player
  MSB_HD MSB_NOINLINE int rem_copy_entity(int e) {
    int c = dup_entity(e);   // same world as e: the one attribute that is re-bound
    if (fault()) return c;
    int L = m.ld8(E_REM + e);
    if (L == REM_LOST) {
      m.st8(E_REM + c, REM_LOST);
    } else if (L != REM_NONE && rem_n(L) > 0) {
      int memo[WORLD_CAP + 1];
      for (int i = 0; i <= WORLD_CAP; i++) memo[i] = -1;
      int d = rem_deep_copy(L, memo, e, c);
      if (fault()) return c;
      m.st8(E_REM + c, d);
    }
    return c;
  }
'''

B[(1269, 1343)] = r'''  // ------------------------------------------------------------------------------------------
  // Control flow: an explicit work stack instead of the reference's recursion
  // ------------------------------------------------------------------------------------------
  // The reference recurses: Unit.move -> activate_ability -> deal_damage -> destroy -> pop_trigger -> activate_ability
  // -> command -> move ... (unit.py:124-231, card.py:48-62, board.py:46-56).  Here every function on such a cycle is a
  // FRAME on the game's work stack (M::sk_ld / sk_st: on the device LDS words next to the record, state.h) and run()
  // is the only loop: it takes the top frame and executes its handler until the handler "calls" -- pushes the callee's
  // frame and returns to run() -- or finishes and pops itself.  A handler is a forward-only state machine: the
  // frame's `state` says where to resume, and a backward jump (the next round of a loop whose body calls) goes through
  // run().  A call that completes without pushing anything (damage that kills nobody and fires nothing) continues
  // inline.  A fault ends the step at once, like the exception it stands for (run() drops the stack).
  //   frame = header word on top {fn, state, a, b: one byte each} + the frame's other words below it
  //   call_X(k, ...): the part of X before its first nested call runs at once, in the caller; what is left of X, if
  //   anything, waits in a frame.  The caller tells "completed" from "pending" by the stack pointer.
  struct Wk {
    int sp;       // words in use on this game's work stack
    int result;   // Stormbound.step's reward | done << 1
  };
  enum : int { F_STEP = 1, F_UNIT_PLAY, F_MOVE, F_RUNAB, F_CTXLEAVE, F_DESTROY_TAIL, F_CMD_TAIL, F_EACH, F_AFTER, F_TURN };
  MSB_HD MSB_INL static uint32_t mk_hdr(int fn, int st, int a, int b) {
    return (uint32_t)(fn & 0xff) | ((uint32_t)(st & 0xff) << 8) | ((uint32_t)(a & 0xff) << 16) | ((uint32_t)(b & 0xff) << 24);
  }
  MSB_HD MSB_INL static int hdr_fn(uint32_t h) { return (int)(h & 0xff); }
  MSB_HD MSB_INL static int hdr_st(uint32_t h) { return (int)((h >> 8) & 0xff); }
  MSB_HD MSB_INL static int hdr_a(uint32_t h) { return (int)((h >> 16) & 0xff); }
  MSB_HD MSB_INL static int hdr_b(uint32_t h) { return (int)(h >> 24); }
  MSB_HD MSB_INL void wk_push(Wk& k, uint32_t v) {
    m.sk_st(k.sp, v);
    k.sp++;
  }
  // a point / slot list as six stack words (the three 64-bit lanes of a PList; the length travels in the frame)
  MSB_HD MSB_INL void wk_push_list(Wk& k, const PList& l) {
    for (int j = 2; j >= 0; j--) {
      wk_push(k, (uint32_t)(l.w[j] >> 32));
      wk_push(k, (uint32_t)l.w[j]);
    }
  }
  MSB_HD MSB_INL PList wk_list(int first, int n) {   // `first` = stack index of the word pushed last (low half of lane 0)
    PList l;
    for (int j = 0; j < 3; j++)
      l.w[j] = (unsigned long long)m.sk_ld(first - 2 * j) | ((unsigned long long)m.sk_ld(first - 2 * j - 1) << 32);
    l.set_n(n);
    return l;
  }
  MSB_HD MSB_INL void wk_store_list(int first, const PList& l) {
    for (int j = 0; j < 3; j++) {
      m.sk_st(first - 2 * j, (uint32_t)l.w[j]);
      m.sk_st(first - 2 * j - 1, (uint32_t)(l.w[j] >> 32));
    }
  }
  // ctx_leave(saved) once everything pushed after it has run: the tail of an entity method that switched worlds
  MSB_HD MSB_INL void wk_push_ctx(Wk& k, int sv) {
    if (REM_LISTS && sv >= 0) wk_push(k, mk_hdr(F_CTXLEAVE, 0, sv, 0));
  }

  // ------------------------------------------------------------------------------------------
  // Deferred triggers: Board.push_trigger / pop_trigger (board.py:46-56) and the wrapper that
  // Card.__init_subclass__ puts around every overridden activate_ability (card.py:48-62).
  // ------------------------------------------------------------------------------------------
  MSB_HD MSB_INL void push_trigger(int e, bool src) {
    int n = m.ld8(H_TRIG_N);
    if (n >= TRIG_CAP) {
      set_fault(FAULT_TRIG_STACK);
      return;
    }
    trig_put(n, e, src);
    m.st8(H_TRIG_N, n + 1);
  }
  // entry i of the trigger stack: the slot in the byte, has_source in its top bit -- or, where slot ids need all eight
  // bits (TRIG_WIDE), in bit i of X_TRIGSRC
  MSB_HD MSB_INL void trig_put(int i, int e, bool src) {
    if (TRIG_WIDE) {
      m.st8(OFF_TRIG + i, e);
      uint32_t f = m.ld32(X_TRIGSRC);
      m.st32(X_TRIGSRC, src ? f | (1u << i) : f & ~(1u << i));
    } else
      m.st8(OFF_TRIG + i, e | (src ? 0x80 : 0));
  }
  MSB_HD MSB_INL int trig_slot(int i) const { return m.ld8(OFF_TRIG + i) & TRIG_SLOT; }
  MSB_HD MSB_INL bool trig_src(int i) const { return TRIG_WIDE ? ((m.ld32(X_TRIGSRC) >> i) & 1u) != 0 : (m.ld8(OFF_TRIG + i) & 0x80) != 0; }
  MSB_HD MSB_INL void call_pop_trigger(Wk& k) {
    int n = m.ld8(H_TRIG_N);
    if (n == 0 || m.ld8(H_RESOLVING)) return;
    int e = trig_slot(n - 1);
    bool src = trig_src(n - 1);
    m.st8(H_TRIG_N, n - 1);
    call_run_ability(k, e, -1, PK_NONE, src);
  }
  // wrapped activate_ability (card.py:48-62).  e >= 0: entity slot; e < 0: a spell, `spell` = card | owner << 8.
  // Frame F_RUNAB {hdr: state, e (0xFF = a spell), recursion depth outside | src << 7; w1: spell | pos_pk << 16}:
  // state 0 = the ability has yet to start, 1 = it has returned.  The ability itself starts on run()'s next turn, so
  // that the card code exists once, in the handler.
  MSB_HD MSB_INL void call_run_ability(Wk& k, int e, int spell, int pos_pk, bool src) {
    const int sv = e >= 0 ? ctx_enter(e) : -1;   // the wrapper works on self.player.board (card.py:54-60)
    if (fault()) return;
    wk_push_ctx(k, sv);
    const int d = m.ld8(H_DEPTH);
    if (d >= MAX_DEPTH || k.sp > SK_CAP - SK_MARGIN) {
      set_fault(FAULT_DEPTH);
      return;
    }
    m.st8(H_DEPTH, d + 1);
    wk_push(k, (uint32_t)(spell & 0xffff) | ((uint32_t)(pos_pk & 0xff) << 16));
    wk_push(k, mk_hdr(F_RUNAB, 0, e, d | (src ? 0x80 : 0)));
  }
  MSB_HD MSB_INL void h_runab(Wk& k, const uint32_t hdr) {
    const int top = k.sp - 1;
    const int d = hdr_b(hdr) & 0x7f;
    if (hdr_st(hdr) == 0) {
      const int e = hdr_a(hdr);
      const bool src = (hdr_b(hdr) & 0x80) != 0;
      const uint32_t w1 = m.sk_ld(top - 1);
      const int pos_pk = (int)((w1 >> 16) & 0xff);
      m.st8(H_RESOLVING, 1);
      m.sk_st(top, hdr | (1u << 8));   // whatever the ability pushes, this frame resumes behind it
      if (e != 0xff) {
        M::trace_ability(e_card(e), m.ld8g(eg(e), EO_POS));   // diagnostics hook: nothing in the product
        ability_entity(k, e, pos_pk, src);
      } else {
        const int spell = (int)(w1 & 0xffff);
        M::trace_ability(spell & 0xff, -1);
        ability_spell(k, spell & 0xff, spell >> 8, pos_pk);
      }
      if (fault() || k.sp - 1 != top) return;   // raised, or still running (its frames are above this one)
    }
    // the ability has returned, card.py:54-60
    m.st8(H_RESOLVING, 0);
    const int n = m.ld8(H_TRIG_N);
    if (n == 0) {
      m.st8(H_DEPTH, d);
      k.sp = top - 1;
      return;
    }
    // the wrapper's trailing pop_trigger() is a tail call in the reference: this frame runs the next deferred ability
    const int e = trig_slot(n - 1);
    const bool src = trig_src(n - 1);
    m.st8(H_TRIG_N, n - 1);
    m.sk_st(top - 1, 0xffffu | ((uint32_t)PK_NONE << 16));
    m.sk_st(top, mk_hdr(F_RUNAB, 0, e, d | (src ? 0x80 : 0)));
  }
  // entity.activate_ability(...) as called by the engine: wrapped iff the class overrides it.
  MSB_HD MSB_INL void call_activate(Wk& k, int e, int pos_pk, bool src) {
    if (e_has_ability(e)) call_run_ability(k, e, -1, pos_pk, src);
  }
'''

B[(1355, 1387)] = r'''  // Unit.deal_damage unit.py:205-219 / Structure.deal_damage structure.py:52-63.  The reference returns the amount
  // dealt; the one caller that uses it (cards/u405.py) takes min(amount, strength) itself (EACH_DMG_HEAL).
  MSB_HD MSB_INL void call_entity_damage(Wk& k, int e, int amount, bool pending, bool src) {
    const int sv = ctx_enter(e);
    if (fault()) return;
    wk_push_ctx(k, sv);
    int s = e_str(e);
    if (s - amount < 0) amount = s;
    e_set_dmg(e, amount);
    s -= amount;
    e_set_str(e, s);
    if (!pending && s <= 0) {
      call_destroy(k, e, src);
    } else if (e_trigger(e) == TR_AFTER_SURVIVING && s > 0) {
      push_trigger(e, src);
      if (fault()) return;
      call_pop_trigger(k);
    }
  }
  // X.deal_damage(amount, source=...) where X = board.at(point): unit, structure or Player
  MSB_HD MSB_INL void call_damage(Wk& k, int who, int amount, bool src) {
    if (who >= AT_PLAYER) {
      player_deal_damage(who - AT_PLAYER, amount);
      return;
    }
    if (who < 0) {
      set_fault(FAULT_PY_EXCEPTION);  // None.deal_damage
      return;
    }
    call_entity_damage(k, who, amount, false, src);
  }'''

B[(1396, 1421)] = r'''  // Unit.destroy unit.py:221-231 / Structure.destroy structure.py:65-69.  Frame F_DESTROY_TAIL: what a unit's destroy
  // does once its ON_DEATH ability has returned.
  MSB_HD MSB_INL void call_destroy(Wk& k, int e, bool src) {
    const int sv = ctx_enter(e);
    if (fault()) return;
    wk_push_ctx(k, sv);
    if (e_is_unit(e)) {
      board_set(e_pos(e), -1);
      m.st8g(eg(e), EO_PATHN, 0);
      e_set_dmg(e, e_str(e));
      if (e_card_trigger(e) == TR_ON_DEATH) {
        push_trigger(e, src);
        if (fault()) return;
        if (!m.ld8(H_RESOLVING)) {   // pop_trigger() runs an ability now: the rest of destroy waits for it
          wk_push(k, mk_hdr(F_DESTROY_TAIL, 0, 0, 0));
          call_pop_trigger(k);
          return;
        }
      }
      recalc_front_after_destroy();
    } else {
      e_set_dmg(e, e_str(e));
      board_set(e_pos(e), -1);
      recalc_front_after_destroy();
    }
  }'''

B[(1532, 1634)] = r'''  // Unit.move, unit.py:124-203.  Frame F_MOVE {hdr: state, e, recursion depth outside; w1: the path list bound when
  // the loop started (fact #5); w2: i | n << 3 | move_id << 8 | target << 16 | flags << 24 (1 is_attacked,
  // 2 target_pending, 4 local_pending); w3: the target's strength before the fight}.  The states are the places where
  // the reference's move() is waiting for a nested call.
  enum : int { MV_ENTRY = 0, MV_POISONED, MV_BEFORE_MOVING, MV_STEP, MV_BASE_HIT, MV_FIGHT, MV_STRUCK, MV_STRUCK_BACK,
               MV_TARGET_DEAD, MV_SELF_DEAD, MV_AFTER_ATTACK, MV_DONE };
  MSB_HD MSB_INL void call_move(Wk& k, int e) {
    const int sv = ctx_enter(e);
    if (fault()) return;
    wk_push_ctx(k, sv);
    const int d = m.ld8(H_DEPTH);
    if (d >= MAX_DEPTH || k.sp > SK_CAP - SK_MARGIN) {
      set_fault(FAULT_DEPTH);
      return;
    }
    m.st8(H_DEPTH, d + 1);
    wk_push(k, 0);
    wk_push(k, 0);
    wk_push(k, 0);
    wk_push(k, mk_hdr(F_MOVE, MV_ENTRY, e, d));
  }
  MSB_HD MSB_INL void h_move(Wk& k, const uint32_t hdr) {
    const int top = k.sp - 1;
    const int e = hdr_a(hdr), d = hdr_b(hdr);
    uint32_t path = m.sk_ld(top - 1);
    const uint32_t w2 = m.sk_ld(top - 2);
    int i = (int)(w2 & 7), n = (int)((w2 >> 3) & 7), current_id = (int)((w2 >> 8) & 0xff), target = (int)((w2 >> 16) & 0xff);
    int flags = (int)(w2 >> 24);
    int cached = (int)(int16_t)(m.sk_ld(top - 3) & 0xffffu);
    const int trig = e_trigger(e);
    int next = MV_DONE;
    P dest{0, 0};
    int owner = 0, tp = 0;
#define MV_WAIT(s_)  \
  do {               \
    next = (s_);     \
    goto wait;       \
  } while (0)
#define MV_CALLED(s_)                        \
  do {                                       \
    if (fault()) return;                     \
    if (k.sp - 1 != top) MV_WAIT(s_);        \
  } while (0)
    switch (hdr_st(hdr)) {
      case MV_ENTRY:
        current_id = (m.ld8g(eg(e), EO_MOVEID) + 1) & 0xff;
        m.st8g(eg(e), EO_MOVEID, current_id);
        if (phase() == PH_TURN_START) {
          if (e_st(e, ST_POISONED) > 0) {
            call_entity_damage(k, e, 1, false, false);
            MV_CALLED(MV_POISONED);
          } else if (e_st(e, ST_VITALIZED) > 0)
            e_set_str(e, e_str(e) + 1);
        }
        // fall through
      case MV_POISONED:
        if (phase() == PH_TURN_START && e_frozen(e)) {
          e_st_remove(e, ST_FROZEN);
          goto done;
        }
        if (m.ld8g(eg(e), EO_PATHN) == 0) goto done;
        if (trig == TR_BEFORE_MOVING && !e_disabled(e)) {
          call_run_ability(k, e, -1, PK_NONE, true);
          if (fault()) return;
          MV_WAIT(MV_BEFORE_MOVING);
        }
        // fall through
      case MV_BEFORE_MOVING:
        if (e_frozen(e)) goto done;
        // `for destination in self.path` iterates the list object bound now (fact #5)
        n = m.ld8g(eg(e), EO_PATHN);
        path = e_path(e);
        i = 0;
        // fall through
      case MV_STEP:
        if (i >= n) goto done;
        dest = p_unpack((path >> (8 * i)) & 0xff);
        owner = e_owner(e);  // self.player is re-read by the reference; convert() may change it
        flags = 0;
        if (dest.y < 0 || dest.y > 4) {
          if (trig == TR_BEFORE_ATTACKING && !e_disabled(e)) {
            call_run_ability(k, e, -1, p_pack(dest), true);
            if (fault()) return;
            MV_WAIT(MV_BASE_HIT);
          }
          goto base_hit;
        }
        target = at(dest);
        if (target != AT_NONE && e_owner(target) == owner && dest.x == e_pos(e).x) goto done;
        if (!(target != AT_NONE && (e_confused(e) || e_owner(target) != owner))) goto advance;
        if (trig == TR_BEFORE_ATTACKING && !e_disabled(e)) {
          call_run_ability(k, e, -1, p_pack(dest), true);
          if (fault()) return;
          MV_WAIT(MV_FIGHT);
        }
        goto fight;
      case MV_BASE_HIT:
        dest = p_unpack((path >> (8 * i)) & 0xff);
      base_hit:
        tp = dest.y < 0 ? remote() : local();
        player_deal_damage(tp, e_str(e));
        if (pl_base(tp) > 0) {
          call_destroy(k, e, false);
          MV_CALLED(MV_DONE);
        }
        goto done;
      case MV_FIGHT:
        dest = p_unpack((path >> (8 * i)) & 0xff);
      fight:
        target = at(dest);
        if (target == AT_NONE) goto advance;
        cached = e_str(target);   // target_strength_cached
        flags = ((e_trigger(target) == TR_ON_DEATH && !e_disabled(target)) ? 2 : 0) | ((trig == TR_ON_DEATH && !e_disabled(e)) ? 4 : 0);
        call_entity_damage(k, target, e_str(e), (flags & 2) != 0, false);
        MV_CALLED(MV_STRUCK);
        // fall through
      case MV_STRUCK:
        call_entity_damage(k, e, cached, (flags & 4) != 0, false);
        MV_CALLED(MV_STRUCK_BACK);
        // fall through
      case MV_STRUCK_BACK:
        if (e_str(target) <= 0 && (flags & 2)) {
          call_destroy(k, target, false);
          MV_CALLED(MV_TARGET_DEAD);
        }
        // fall through
      case MV_TARGET_DEAD:
        if (e_str(e) <= 0 && (flags & 4)) {
          call_destroy(k, e, false);
          MV_CALLED(MV_SELF_DEAD);
        }
        // fall through
      case MV_SELF_DEAD:
        flags |= 1;   // is_attacked
      advance:
        dest = p_unpack((path >> (8 * i)) & 0xff);
        if (current_id != m.ld8g(eg(e), EO_MOVEID)) goto done;
        if (at(dest) == AT_NONE && e_str(e) > 0) {
          board_set(e_pos(e), -1);
          board_set(dest, e);
          const int o = e_owner(e);
          if (pl_front(o) > dest.y) set_pl_front(o, dest.y > 1 ? dest.y : 1);
          if ((flags & 1) && trig == TR_AFTER_ATTACKING && !e_disabled(e)) {
            call_run_ability(k, e, -1, PK_NONE, true);
            if (fault()) return;
            MV_WAIT(MV_AFTER_ATTACK);
          }
          if (e_confused(e)) e_st_remove(e, ST_CONFUSED);
        }
        goto next_step;
      case MV_AFTER_ATTACK:
        if (e_confused(e)) e_st_remove(e, ST_CONFUSED);
      next_step:
        i++;
        if (i < n) MV_WAIT(MV_STEP);   // the next destination: a backward jump, through run()
        goto done;
      default:   // MV_DONE
        break;
    }
  done:
    m.st8(H_DEPTH, d);
    k.sp = top - 3;
    return;
  wait:
    m.sk_st(top - 1, path);
    m.sk_st(top - 2, (uint32_t)(i & 7) | ((uint32_t)(n & 7) << 3) | ((uint32_t)(current_id & 0xff) << 8) | ((uint32_t)(target & 0xff) << 16) |
                         ((uint32_t)(flags & 0xff) << 24));
    m.sk_st(top - 3, (uint32_t)(cached & 0xffff));
    m.sk_st(top, mk_hdr(F_MOVE, next, e, d));
#undef MV_WAIT
#undef MV_CALLED
  }

  // Unit.play, unit.py:66-76.  Frame F_UNIT_PLAY {hdr: state, e}: state 0 = the ON_PLAY ability has returned, 1 = move() has.
  MSB_HD MSB_INL void call_unit_play(Wk& k, int e, P position) {
    e_set_flag(e, EF_RESOLVING_PLAY, true);
    board_set(position, e);
    set_path(e, true);
    if (fault()) return;
    if (e_card_trigger(e) == TR_ON_PLAY) {
      wk_push(k, mk_hdr(F_UNIT_PLAY, 0, e, 0));
      call_run_ability(k, e, -1, PK_NONE, true);
    } else {
      wk_push(k, mk_hdr(F_UNIT_PLAY, 1, e, 0));
      call_move(k, e);
    }
  }
  MSB_HD MSB_INL void h_unit_play(Wk& k, const uint32_t hdr) {
    const int top = k.sp - 1, e = hdr_a(hdr);
    if (hdr_st(hdr) == 0) {
      m.sk_st(top, mk_hdr(F_UNIT_PLAY, 1, e, 0));
      call_move(k, e);
      return;
    }
    e_set_flag(e, EF_RESOLVING_PLAY, false);
    k.sp = top;
  }
  // Structure.play, structure.py:45-50
  MSB_HD MSB_INL void call_structure_play(Wk& k, int e, P position) {
    board_set(position, e);
    if (e_card_trigger(e) == TR_ON_PLAY) call_run_ability(k, e, -1, PK_NONE, true);
  }'''

B[(1643, 1657)] = r'''  // Unit.command, unit.py:282-289.  Frame F_CMD_TAIL {hdr: e, fixedly_forward as it was}: behind the move.
  MSB_HD MSB_INL void call_command(Wk& k, int e) {
    if (!need_unit(e)) return;
    const int sv = ctx_enter(e);
    if (fault()) return;
    wk_push_ctx(k, sv);
    const bool ff = e_ff(e);
    e_set_flag(e, EF_FF, true);
    set_path(e, false);
    if (fault()) return;
    wk_push(k, mk_hdr(F_CMD_TAIL, 0, e, ff ? 1 : 0));
    call_move(k, e);
  }'''

B[(1713, 1745)] = r'''  // Unit.force_attack, unit.py:341-371 (the move at its end is a tail call)
  MSB_HD MSB_INL void call_force_attack(Wk& k, int e, P dest) {
    const int sv = ctx_enter(e);
    if (fault()) return;
    wk_push_ctx(k, sv);
    P pos = e_pos(e);
    if ((dest.x != pos.x && dest.y != pos.y) || at(dest) == AT_NONE) return;
    bool vertical = dest.x == pos.x;
    int fixed = vertical ? pos.x : pos.y;
    int start = vertical ? pos.y : pos.x;
    int end = vertical ? dest.y : dest.x;
    int delta = end > start ? 1 : -1;
    uint32_t packed = 0;
    int n = 0;
    for (int i = start + delta; i != end + delta; i += delta) {
      P pt = vertical ? P{fixed, i} : P{i, fixed};
      if (i != end && at(pt) != AT_NONE) return;
      if (n < PATH_CAP) packed |= (uint32_t)p_pack(pt) << (8 * n);
      n++;
    }
    if (n > PATH_CAP) {
      set_fault(FAULT_CAP_PATH);
      return;
    }
    if (n > 0) {
      e_set_path(e, packed);
      m.st8g(eg(e), EO_PATHN, n);
      call_move(k, e);
    }
  }'''

B[(1886, 1937)] = r'''  // Player.play, player.py:68-77.  has_pos=false <=> position None
  MSB_HD MSB_INL void call_player_play(Wk& k, int o, int index, P position, bool has_pos) {
    int card = hand_card(o, index), fl = hand_flags(o, index);
    int strength = inst_strength(card, fl, hand_x(o, index));   // target.copy() copies the instance's strength
    add_history(o, card);
    discard(o, index);
    if (fault()) return;
    const CardInfo& ci = g_cards[card];
    if (ci.kind == KIND_UNIT) {
      int e = new_entity(card, o, strength, ci.movement, (fl & CF_FF) != 0);   // target.copy()
      if (fault()) return;
      if (!has_pos) {
        set_fault(FAULT_PY_EXCEPTION);
        return;
      }
      call_unit_play(k, e, position);
    } else if (ci.kind == KIND_STRUCT) {
      int e = new_entity(card, o, strength, 0, false);
      if (fault()) return;
      if (fl & CF_SINGLE_USE) e_set_flag(e, EF_SINGLE_USE, true);
      if (!has_pos) {
        set_fault(FAULT_PY_EXCEPTION);
        return;
      }
      call_structure_play(k, e, position);
    } else {
      call_spell_play(k, card, o, position, has_pos);
    }
  }
  // Spell.play, spell.py:22-24
  MSB_HD MSB_INL void call_spell_play(Wk& k, int card, int o, P position, bool has_pos) {
    const CardInfo& ci = g_cards[card];
    bool go = true;
    if (ci.tgt.has) {
      Tgt t = mk_tgt(ci.tgt);
      PList l = get_targets(cp(), t, PK_NONE);
      go = has_pos && l.has(position);
      // `None in [Point...]` evaluates Point.__eq__(None) -> AttributeError when the list is non-empty
      if (!has_pos && l.n() > 0) {
        set_fault(FAULT_PY_EXCEPTION);
        return;
      }
    }
    if (go) call_run_ability(k, -1, card | (o << 8), has_pos ? p_pack(position) : PK_NONE, true);
  }'''

B[(1971, 2011)] = r'''  // Board.to_next_turn, board.py:117-145.  Frame F_TURN {hdr: state, i, ns; six words: the snapshot of entity OBJECTS
  // being iterated (fact #6), as slot ids}: state 0 = the friendly structures' TURN_START abilities, 1 = the units' moves.
  MSB_HD MSB_INL void call_next_turn(Wk& k) {
    m.st8(H_PHASE, PH_TURN_END);
    int ender = cp();
    fill_hand(ender);
    if (fault()) return;
    // No card in the reference has a TURN_END trigger (structure.py:8 default is [TURN_START],
    // b305 is [ON_PLAY]); the TURN_END loop (board.py:121-123) never fires an ability.
    calculate_front_line(local());
    calculate_front_line(remote());
    m.st16(pl(ender, P_MAXMANA), pl_maxmana(ender) + 1);
    set_pl_mana(0, pl_maxmana(0));
    set_pl_mana(1, pl_maxmana(1));
    m.st8(H_PHASE, PH_TURN_START);
    int ncp = (ender == local()) ? remote() : local();
    m.st8(H_CP, ncp);
    m.st8(pl(ncp, P_FLAGS), m.ld8(pl(ncp, P_FLAGS)) | 3);
    PList hs;
    hs.clear();
    PList snap = get_targets(ncp, mk_tgt(TK_STRUCTURE, TS_FRIENDLY), PK_NONE);
    int ns = snap.n();
    for (int i = 0; i < ns; i++) hs.set8(i, at(snap.at(i)));
    wk_push_list(k, hs);
    wk_push(k, mk_hdr(F_TURN, 0, 0, ns));
  }
  MSB_HD MSB_INL void h_turn(Wk& k, const uint32_t hdr) {
    const int top = k.sp - 1;
    int i = hdr_a(hdr), ns = hdr_b(hdr);
    PList hs = wk_list(top - 1, 0);
    if (hdr_st(hdr) == 0) {
      while (i < ns) {
        const int s = hs.get8(i);
        i++;
        // structure.is_at_turn_start: token structures and b001 run the empty base ability
        if (e_card_trigger(s) == TR_TURN_START) {
          m.sk_st(top, mk_hdr(F_TURN, 0, i, ns));
          call_run_ability(k, s, -1, m.ld8g(eg(s), EO_POS) /*unused*/, true);
          return;
        }
      }
      // the units as they stand once every structure has acted
      PList snap = get_targets(cp(), mk_tgt(TK_UNIT, TS_FRIENDLY), PK_NONE);
      ns = snap.n();
      for (int j = 0; j < ns; j++) hs.set8(j, at(snap.at(j)));
      wk_store_list(top - 1, hs);
      i = 0;
    }
    if (i < ns) {
      const int u = hs.get8(i);
      m.sk_st(top, mk_hdr(F_TURN, 1, i + 1, ns));
      set_path(u, false);
      if (fault()) return;
      call_move(k, u);
      return;
    }
    m.st8(H_PHASE, PH_PLAY);
    k.sp = top - 6;
  }
'''

B[(2079, 2132)] = r'''  // Stormbound.step, games/stormbound.py:318-373 (without the observation; see observe.inc).
  // The caller guarantees `action` is in legal_actions().  Returns reward | done << 1 as the reference
  // computes them.  Frame F_STEP {hdr: state, action}: state 0 = the card has been played, 1 = the turn has been passed on.
  MSB_HD MSB_INL int step(int action) {
    Wk k{0, 0};
    begin_step();
    if (fault()) return 0;
    int lo = local();
    wk_push(k, mk_hdr(F_STEP, 0, action, 0));
    if (action < 148) {
      // PLACE: card = a//16, tile = a%16 over y=4..1,x=0..3.  USE: card = (a-64)//21, idx = (a-64)%21; the
      // countdown executes at the idx-th tile of y=4..0,x=0..3 -- one tile after the one Action.to_int
      // encoded (fact #2); idx==20 falls off the loop: nothing happens at all.
      bool place = action < 64;
      int ci = place ? action >> 4 : (action - 64) / 21;
      int idx = place ? action & 15 : (action - 64) % 21;
      if (idx < 20) {
        P pos{idx & 3, 4 - (idx >> 2)};
        bool has_pos = place || g_cards[hand_card(lo, ci)].tgt.has != 0;
        set_pl_mana(lo, pl_mana(lo) - hand_cost(lo, ci));
        call_player_play(k, lo, ci, pos, has_pos);
      }
    } else if (action < 152) {
      cycle(lo, action - 148);
      m.st8(pl(lo, P_FLAGS), m.ld8(pl(lo, P_FLAGS)) & ~1);
    } else if (action < 155) {
      int ci = action - 151;
#if defined(MSB_EXT) && MSB_EXT
      int a = hand_id(lo, ci), b = hand_id(lo, 0);
      m.st8(pl(lo, P_HAND + ci), b);
      m.st8(pl(lo, P_HAND), a);
#else
      uint32_t a = m.ld32(pl(lo, P_HAND + 4 * ci)), b = m.ld32(pl(lo, P_HAND));
      m.st32(pl(lo, P_HAND + 4 * ci), b);
      m.st32(pl(lo, P_HAND), a);
#endif
      m.st8(pl(lo, P_FLAGS), m.ld8(pl(lo, P_FLAGS)) & ~2);
    }
    run(k);
    return k.result;   // 0 if the play raised; what was computed before the turn was passed on otherwise
  }
  MSB_HD MSB_INL void h_step(Wk& k, const uint32_t hdr) {
    const int top = k.sp - 1, action = hdr_a(hdr);
    if (hdr_st(hdr) == 0) {
      // done = have_winner() or len(legal_actions()) == 0; legal_actions() is never empty (PASS)
      k.result = (pl_base(remote()) <= 0 ? 1 : 0) | (have_winner() ? 2 : 0);
      if (action == 155) {
        m.st8(H_TOPLAY, local() ^ 1);
        flip();
        m.sk_st(top, mk_hdr(F_STEP, 1, action, 0));
        call_next_turn(k);
        return;
      }
    }
    if (m.ld8(H_RNGOVER)) set_fault(FAULT_RNG_OVERRUN);
    k.sp = top;
  }

  // The handlers of the remaining frames, and run(): the only loop of a step.
  MSB_HD MSB_INL void h_ctx_leave(Wk& k, const uint32_t hdr) {
    ctx_leave(hdr_a(hdr));
    k.sp--;
  }
  MSB_HD MSB_INL void h_destroy_tail(Wk& k, const uint32_t) {
    recalc_front_after_destroy();
    k.sp--;
  }
  MSB_HD MSB_INL void h_cmd_tail(Wk& k, const uint32_t hdr) {
    e_set_flag(hdr_a(hdr), EF_FF, hdr_b(hdr) != 0);
    k.sp--;
  }
  MSB_HD MSB_INL void wk_dispatch(Wk& k, int fn, const uint32_t hdr) {
    switch (fn) {
      case F_STEP: h_step(k, hdr); break;
      case F_UNIT_PLAY: h_unit_play(k, hdr); break;
      case F_MOVE: h_move(k, hdr); break;
      case F_RUNAB: h_runab(k, hdr); break;
      case F_CTXLEAVE: h_ctx_leave(k, hdr); break;
      case F_DESTROY_TAIL: h_destroy_tail(k, hdr); break;
      case F_CMD_TAIL: h_cmd_tail(k, hdr); break;
      case F_EACH: h_each(k, hdr); break;
      case F_AFTER: h_after(k, hdr); break;
      case F_TURN: h_turn(k, hdr); break;
      default: set_fault(FAULT_UNSUPPORTED); break;   // not a frame: cannot happen
    }
  }
  // Run the work stack until it is empty.  On the device the lanes of a wave are grouped by the function of their top
  // frame first (a "waterfall": take the first waiting lane's function as a wave-uniform value, serve the lanes
  // holding it, repeat), so the switch runs on a scalar and lanes that are in the same function -- whatever path of
  // calls took them there -- execute it together.
  MSB_HD MSB_INL void run(Wk& k) {
    while (k.sp > 0 && !fault()) {
      const uint32_t hdr = m.sk_ld(k.sp - 1);
      const int fn = hdr_fn(hdr);
#if defined(__HIP_DEVICE_COMPILE__)
      int fv = fn;
      asm volatile("" : "+v"(fv));   // an opaque copy: under `fn == f0` the compiler would switch on the vector `fn` again
      for (unsigned long long todo = __ballot(1); todo;) {
        const int leader = __builtin_ctzll(todo);
        const int f0 = __builtin_amdgcn_readlane(fn, leader), f1 = __builtin_amdgcn_readlane(fv, leader);
        const bool mine = fn == f0;
        todo &= ~__ballot(mine);
        if (mine) wk_dispatch(k, f1, hdr);
      }
#else
      wk_dispatch(k, fn, hdr);
#endif
    }
  }
'''

# apply bottom-up
for (a, b) in sorted(B.keys(), reverse=True):
    lines[a - 1:b] = B[(a, b)].split("\n")
out = "\n".join(lines)
# the function-scope profiling hooks are gone with the functions they timed
out = re.sub(r"^\s*MSB_(PRECALL|POSTCALL|SCOPE)\([^)]*\);\s*\n", "", out, flags=re.M)
open(DST, "w").write(out)
print("ok", len(out.split("\n")))
