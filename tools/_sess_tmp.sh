cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python tools/debug_shard.py relax_ds_sh 4 1 30 1 2>&1 | tail -12
