cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in asm as4; do echo "== $v"; HAGRID_AMD_LIB=$GRAFT_REPO_ROOT/tools/_ab/lib_$v.so timeout 600 python tools/dev_traverse_time.py 2>/dev/null | cut -c1-100; done; done
