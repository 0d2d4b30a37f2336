import sys, os, faulthandler
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import scenario_lib as S
from monsoon_amd.cards import CARD_INDEX
from monsoon_amd.engine import BatchEngine
ext_cards = [CARD_INDEX["ua20"], CARD_INDEX["b005"]]
engs = {False: BatchEngine(2), True: BatchEngine(2, extended=True)}
n = 0
for case in S.load():
    for k, rec in enumerate(case["records"]):
        ext = S.needs_extended(rec, ext_cards)
        eng = engs[ext]
        st = rec["before"]
        print(case["test"], k, rec["op"], "ext" if ext else "std", "state ints", len(S.encode_state(st)), flush=True)
        fb = eng.debug_build(0, st["seed"], st["stream_pos"], S.encode_state(st))
        print("  build fault", fb, flush=True)
        f, log = eng.debug_op(0, S.encode_op(rec))
        print("  op fault", f, "log", log[:4], "ok" if eng.export(0).hex() == rec["after"] else "STATE DIFFERS", flush=True)
        n += 1
        if n >= int(sys.argv[1]):
            sys.exit(0)
