python -m pytest tests -m gpu -x -q -k "mlp_launch or forward_matches or coalesced or full_size_reference or options_are or fc_matches" 2>&1 | tail -3
bash tools/_run17.sh 2>&1 | grep -E "^==|stream_kernel|gemm_kernel"
