S="--no-variants --no-sweep --c5-rows 0 --encode-chunks 0 --bm25-docs 0 --hybrid-docs 0 --cpu-rows 0 --rows 1250000 --streams 1 --steps 200 --warmup 20"
run() { python3 bench.py $S 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['sieve'])"; }
run base
MIR_SIEVE_TWO_PHASE_TILES=100000 run one_launch_tpw4
MIR_SIEVE_TWO_PHASE_TILES=100000 MIR_SIEVE_SAMPLE_TPW=8 run one_launch_tpw8
MIR_SIEVE_TWO_PHASE_TILES=100000 MIR_SIEVE_SAMPLE_TPW=16 run one_launch_tpw16
MIR_SIEVE_FIRST_DIV=32 run first_div32
run base
