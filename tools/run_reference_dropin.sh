#!/usr/bin/env bash
# The 30-second test of INTEGRATION.md's drop-in claim, for whoever has BOTH trees on an MI355X (no box of this project's pool does):
#   tools/run_reference_dropin.sh /path/to/InternEvo
# builds libinternevo_hip.so if needed and runs the UNMODIFIED reference's training loop through internevo_amd.plugin.install() for three
# steps of the tiny config, against the reference's own CPU run of the same model (tests/golden/train_cfg0_bf16.json).
set -euo pipefail
here="$(cd "$(dirname "$0")/.." && pwd)"
ref="${1:?usage: $0 <InternEvo checkout>}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$here"
python -c "import __graft_entry__ as g; g.build()"
exec python tools/reference_dropin_check.py "$ref"
