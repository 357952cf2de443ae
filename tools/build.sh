#!/bin/bash
# rebuild libaaerec_hip.so from any working directory
cd "$(dirname "$0")/.." && python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "error|built"
