cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH="$PWD"; timeout 600 python -m pytest tests -m gpu -q -x -k "error_and" 2>&1 | tail -8
