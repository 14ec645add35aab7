set -e
cd $GRAFT_REPO_ROOT
for r in 32 64 16; do echo "stem rows $r"; timeout -k 10 200 python tools/launch_times.py --planes 3 --stem-rows $r | grep "stem\|sum"; done
