#!/bin/bash
# One-factor learning ablation on the CPU oracle loop (scripts/learning_ablation.py): every (flip, 5-key block) cell is one
# single-threaded process, WORKERS of them at a time.  Appends JSON lines to $OUT (default profiles/r03_learning_ablation.jsonl).
#   scripts/run_learning_ablation.sh sac 10         # all SAC flips over keys 0..9
#   scripts/run_learning_ablation.sh ppo 6 1000000  # all PPO flips over keys 0..5 at the reference's num_timesteps
cd "$(dirname "$0")/.."
ALGO=${1:-sac}; NKEYS=${2:-10}; STEPS=${3:-1000000}
OUT=${OUT:-profiles/r03_learning_ablation.jsonl}
WORKERS=${WORKERS:-6}
BLOCK=${BLOCK:-5}
FLIPS=${FLIPS:-$(python scripts/learning_ablation.py list | grep "^$ALGO:" | cut -d: -f2)}
for flip in $FLIPS; do
  for ((k = 0; k < NKEYS; k += BLOCK)); do
    n=$((NKEYS - k < BLOCK ? NKEYS - k : BLOCK))
    echo "$flip $k $n"
  done
done | OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 xargs -P "$WORKERS" -L 1 bash -c \
  'nice -n 10 python scripts/learning_ablation.py '"$ALGO"' $0 $1 $2 '"$STEPS"' >> '"$OUT"' 2>> '"$OUT"'.err'
