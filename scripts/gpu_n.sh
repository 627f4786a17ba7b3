#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python __graft_entry__.py smoke > $OUT/n.smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python -m pytest tests -m gpu -q --maxfail=5 -k "play_cli or distribution_helpers or test_abi or env_step" > $OUT/n.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 8 $OUT/n.pytest.log | cut -c1-300
timeout 300 python play.py --agent_type ValueSim --mcts_sims 50 --max_moves 5 --n_games 4 2>&1 | tail -n 2
