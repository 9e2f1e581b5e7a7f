#!/bin/bash
# Run a command with an experiment build (python-soxr_amd/_variants/<name>/, made by HIPSOXR_VARIANT=<name> build.sh)
# swapped in for the product library, then put the product library back.  GPU box only (the snapshot is scratch).
#   tools/with_variant.sh <name> <command...>
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$R/python-soxr_amd/_variants/$1; shift
P=$R/python-soxr_amd/soxr_amd
[ -f "$V/libhipsoxr.so" ] || { echo "no such variant: $V"; exit 2; }
[ -f "$P/libhipsoxr.so.product" ] || cp "$P/libhipsoxr.so" "$P/libhipsoxr.so.product"
cp "$V/libhipsoxr.so" "$P/libhipsoxr.so"
"$@"; rc=$?
cp "$P/libhipsoxr.so.product" "$P/libhipsoxr.so"
exit $rc
