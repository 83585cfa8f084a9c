#!/bin/bash
# Starts the round-end measurement session on a GPU box FROM A CLEAN, COMMITTED TREE and copies its files to profiles/:
#   bash tools/run_round_end.sh <round tag, e.g. r06> [gpurun timeout s]
# Refuses when `git status --porcelain` shows tracked changes (the tracked profiles must describe a commit, not a work
# tree) or when the library is older than a source file.  The session's SESSION.txt names the commit and the sha256 of the
# library and of bench.py it measured.
set -e
cd "$(dirname "$0")/.."
TAG="${1:?round tag}"
if [ -n "$(git status --porcelain --untracked-files=no)" ]; then echo "tracked files differ from HEAD: commit first" >&2; exit 2; fi
make -q -C gpu-ntt_amd/csrc || { echo "library is stale: make -C gpu-ntt_amd/csrc -j8" >&2; exit 2; }
HEAD_SHA="$(git rev-parse HEAD)"
OUT="gpurun_out/round_end_$TAG"
gpurun --timeout "${2:-3300}" -- "SESSION_HEAD=$HEAD_SHA bash tools/round_end_session.sh $OUT"
grep -q "head $HEAD_SHA" "$OUT/SESSION.txt" || { echo "session did not record this commit" >&2; exit 3; }
for f in "$OUT"/*; do
  b="$(basename "$f")"
  case "$b" in *.err) continue;; esac
  cp "$f" "profiles/${TAG}_$b"
done
echo "copied to profiles/${TAG}_*  (commit $HEAD_SHA)"
