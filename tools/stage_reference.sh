#!/bin/bash
# Stage a TRANSIENT, git-ignored copy of the reference's python tree under oracle/_ref/reference_tree so that ONE gpurun call can run the
# unmodified train_syn.py on a real MI355X (/root/reference does not exist on the GPU box).  Test infrastructure only: the copy is never
# committed (oracle/_ref/ is in .gitignore) and is removed again with `tools/stage_reference.sh --clean` right after the call.
set -e
cd "$(dirname "$0")/.."
D=oracle/_ref/reference_tree
if [ "$1" = "--clean" ]; then rm -rf "$D"; echo "removed $D"; exit 0; fi
REF=${1:-/root/reference}
rm -rf "$D"; mkdir -p "$D"
( cd "$REF" && tar cf - --exclude=imgs --exclude=EMoR --exclude=.git . ) | ( cd "$D" && tar xf - )
du -sh "$D"
