# the constant table's lines touched at the top of the kernel (warm) vs only the stabilisation goal fetched early (xgearly) vs neither (r03)
T="timeout -k 10 300 python tools/tune.py --rounds 3 --arms r03,xgearly,warm,warm_noxg --steps 1000"
G='task_info={stabilization_goal: [0, 0, 1], stabilization_goal_tolerance: 0.05}'
run() { name="$1"; shift; echo "== $name"; $T -- "$@" 2>&1 | grep -A5 "^----" | tail -4; }
run tracking_3d
run stab_3d --env-kw task=stabilization --env-kw "$G"
run info_tier --env-kw info_level=full
run c2_2p16 --workload c2 --envs-per-gpu 65536
run c3_2p18 --workload c3 --envs-per-gpu 262144
run rl_goal_horizon --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
