#!/bin/bash
# The side measurements quoted in DESIGN.md / profiles/README.md, one JSON line each (tools/side_measurements.sh > profiles/rNN_side_measurements.jsonl)
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1; }
run                                   # the metric's configuration (batch 256, frame mode, hop groups of four: whole Griffin-Lim chains per launch)
run --group 2                         # groups of two hops
run --group 0                         # one launch per hop: the deep pipe of round 3 at its default depth 4 (chain segments parked between launches)
run --group 0 --depth 1               # one hop in flight: output after the next submit (wavefront per column + head start)
run --group 0 --depth 2
run --group 0 --depth 3
run --serial                          # one launch per hop, no added latency
run --conv bf16                       # BASELINE config 3
run --stream --graph --batch 1024     # BASELINE config 5 under hipGraph replay
run --stream --graph --graph-hops 8 --batch 1024 --steps 400      # ... eight pushes per captured graph
run --batch 512
run --batch 768
run --batch 1024
run --batch 1024 --queues 2 --depth 2           # two independent pipes of 512 on two HIP streams, split hops
run --batch 2048
run --batch 2048 --queues 2 --depth 1
run --batch 4096
run --batch 8192
run --batch 4096 --queues 2 --pipes 4 --depth 1     # pipes of 1,024 streams taking turns on two HIP streams
run --batch 8192 --queues 2 --pipes 8 --depth 1
run --stream --batch 256 --depth 4
run --stream --batch 256 --group 4                    # streaming hop groups: four hops in, four out per launch
run --stream --graph --batch 256 --group 4            # ... one captured group push replayed
run --batch 384
run --stream --depth 1 --batch 256                    # (device-fed at depth 1 and at 1,024 streams without the graph: what the host-fed lines compare with)
run --stream --batch 1024
run --stream --pcie --batch 256 --steps 2000          # host-fed (zero copy from page-locked memory; the emitted hop leaves with the next launch)
run --stream --pcie --batch 256 --depth 4 --steps 2000
run --stream --pcie --batch 1024 --steps 1000
DN_HOST_DIRECT=1 run --stream --pcie --batch 1024 --steps 1000     # ... stored straight to host memory by its own launch
DN_HOST_STAGED=1 run --stream --pcie --batch 1024 --steps 1000     # ... through staging buffers on two copy queues
run --preset R1
run --preset R1 --batch 1024
run --preset R2
run --pcie
