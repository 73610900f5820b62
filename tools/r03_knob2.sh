cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python tools/robustness_matrix.py 2>&1 | grep -v amdgpu.ids | tail -20
timeout 600 python tools/soak.py 1000 2>&1 | grep -v amdgpu.ids | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
