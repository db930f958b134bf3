cd $GRAFT_REPO_ROOT
export GRAPHLILY_BFS_DEBUG=1
python scripts/r02_bfs_loop.py pokec 2>&1 | grep -v "^pokec device_loop=0\|bits=0" | tail -40
python scripts/r02_bfs_loop.py googleplus 2>&1 | grep BFS_DEBUG | tail -4
