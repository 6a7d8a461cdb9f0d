"""mspa.pipeline.RecordSpill: rank 0's record collector gives the same final order -- canonical sort, then the seeded shuffle --
whether a file's records fit in memory or go through sorted run files and the external merge."""
import json
import os
import random

from mspa import pipeline


def _packed(records_by_name):
    return pipeline._pack_outputs(records_by_name)


def test_external_merge_equals_in_memory(tmp_path):
    rng = random.Random(5)
    batches = []
    for b in range(40):                                      # 40 "scenes", ids colliding across batches, lines of uneven length
        recs = [{"id": f"s{rng.randrange(30):03d}_{rng.randrange(50)}", "v": [b, k], "t": "x" * rng.randrange(0, 200)} for k in range(25)]
        other = [{"id": k + 100 * b, "text": "caf\u00e9 \n two"} for k in range(3)]
        batches.append({"head_a": recs, "head_b": other} if b % 3 else {"head_a": recs})
    out = {}
    for limit in (1 << 30, 4096, 300):                       # all in memory; a few runs; a run per batch and many second-stage chunks
        sp = pipeline.RecordSpill(str(tmp_path / f"spill{limit}"), limit)
        sp.touch("head_empty")
        for b in batches:
            sp.add_packed(_packed(b))
        assert sp.counts == {"head_a": 1000, "head_b": 39, "head_empty": 0} or sp.counts["head_b"] == 3 * sum(1 for i in range(40) if i % 3)
        lines = {name: list(sp.finish(name, random.Random(f"7:{name}"))) for name in sp.names()}
        out[limit] = lines
        assert len(lines["head_a"]) == 1000 and lines["head_empty"] == []
        if limit < 1 << 30:
            assert sp.runs_written > 2
            assert not os.path.isdir(sp.dir) or os.listdir(sp.dir) == []      # run files are removed as they are merged
    assert out[1 << 30] == out[4096] == out[300]
    # and that order IS sort-by-(key, line) followed by random.Random(seed).shuffle
    want = sorted((str(r["id"]), json.dumps(r).encode()) for b in batches for r in b["head_a"])
    random.Random("7:head_a").shuffle(want)
    assert [ln for _k, ln in want] == out[300]["head_a"]
