#!/bin/bash
# Keeps build/head/ = `git archive HEAD` + its freshly built libraries, swapped in atomically.  The GPU poller (poll.sh) runs the
# round's session from THAT tree, so a gpurun call accepted in the middle of an edit still measures a committed state.
set -eu
cd /root/repo
rm -rf build/head.new && mkdir -p build/head.new
git archive HEAD | tar -x -C build/head.new
git rev-parse HEAD > build/head.new/HEAD_SHA
( cd build/head.new && python -c "import __graft_entry__ as g; g.build()" > build.log 2>&1 )
rm -rf build/head.old; [ -d build/head ] && mv build/head build/head.old; mv build/head.new build/head; rm -rf build/head.old
echo "build/head = $(cat build/head/HEAD_SHA)"
