python tools/exp_conv_env.py base 2>&1 | tail -1
MIOPEN_DEBUG_CONV_WINOGRAD=0 python tools/exp_conv_env.py no_winograd 2>&1 | tail -1
MIOPEN_DEBUG_CONV_WINOGRAD=0 BENCHMARK=1 python tools/exp_conv_env.py no_winograd_find 2>&1 | tail -1
MIOPEN_DEBUG_CONV_WINOGRAD=0 MIOPEN_DEBUG_CONV_DIRECT=0 BENCHMARK=1 python tools/exp_conv_env.py igemm_only_find 2>&1 | tail -1
MIOPEN_FIND_MODE=1 BENCHMARK=1 python tools/exp_conv_env.py find_mode_normal 2>&1 | tail -1
MIOPEN_FIND_ENFORCE=3 MIOPEN_USER_DB_PATH=/tmp/miopen BENCHMARK=1 timeout 300 python tools/exp_conv_env.py find_enforce_search 2>&1 | tail -1
