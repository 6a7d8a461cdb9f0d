#!/usr/bin/env python3
"""Prints tools/dropin_ranks.py's JSON as a table: python tools/show_ranks.py file.json [...]"""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print(f"# {f}: {d['scenes']} scenes x {d['frames_per_scene']} frames, {d['num_workers_per_rank']} decode threads x {d.get('scenes_in_flight_per_rank')} scenes in flight, "
          f"{d.get('cpus_per_rank')} CPUs per rank; depth decode: {d.get('depth_decode')}; frames: {d.get('frames', 'noisy')}; host {d.get('host_cpus')}")
    for w, v in d["worlds"].items():
        for n, l in v.items():
            if isinstance(l, dict) and "scenes_per_s" in l:
                r = lambda x: [round(y, 3) if y is not None else None for y in x]
                print(f"  ranks {w} {n.split('.')[0]:32s} {l['scenes_per_s']:7.2f} scenes/s  x{l.get('speedup_vs_1', 1.0):<5} identical={l.get('files_identical_to_1_rank')} "
                      f"passes {l['passes_s']}\n      rank0 consume {l['rank0_consume_busy_s']} drain {l['rank0_writer_drain_s']} backpressure {l['rank0_writer_backpressure_s']}; "
                      f"wait_at_exchange {r(l['wait_at_exchange_s'])}; decode busy {r(l['decode_busy_s'])}; produce {r(l['produce_s'])}; encode {r(l.get('encode_s') or [])} (waited {r(l.get('encode_wait_s') or [])}); stage {r(l.get('stage_s') or [])}; exchange {r(l['exchange_s'])}")
            else:
                print(f"  ranks {w} {n}: {l}")
