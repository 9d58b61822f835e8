#!/bin/bash
# First contact with an 8-GPU node: see scripts/first_contact_8gpu.py (this wrapper only fixes the environment the
# hosts this was built on need and keeps RCCL's INFO log out of the terminal).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=${TMPDIR:-/tmp}
: "${HSA_ENABLE_IPC_MODE_LEGACY:=0}"; export HSA_ENABLE_IPC_MODE_LEGACY
exec python scripts/first_contact_8gpu.py "$@"
