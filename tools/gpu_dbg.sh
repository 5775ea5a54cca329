cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
which rocgdb gdb
for i in 1 2 3 4 5 6; do
  timeout 300 python -X faulthandler -m pytest tests/test_traverse_gpu.py -x -q -k "ray_binning_gives" > gpurun_out/dbg_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc"
  if [ $rc -ne 0 ]; then break; fi
done
GDB=$(which rocgdb || which gdb)
timeout 600 $GDB -batch -ex "handle SIGFPE stop print" -ex run -ex bt -ex "info registers rip" --args python -m pytest tests/test_traverse_gpu.py -x -q -k "ray_binning_gives or tile_order or tail_mode" > gpurun_out/dbg_gdb.log 2>&1
tail -60 gpurun_out/dbg_gdb.log
