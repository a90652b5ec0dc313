#!/bin/bash
# exit cost of a HIP process (GPU box): bash tools/exit_probe.sh   (tools/exit_probe.hip -> bin/exit_probe)
t() { sleep 3; OUT=$("$@"); E=$(date +%s.%N); X=$(echo "$OUT" | sed 's/.*exit_at //'); echo "$* : $OUT -> teardown $(python3 -c "print(round($E - $X, 3))") s"; }
t bin/exit_probe 0 0 0
t bin/exit_probe 0 0 0
t bin/exit_probe 12 0 0
t bin/exit_probe 12 1 0
t bin/exit_probe 9 1 0
t bin/exit_probe 0 0 3
t bin/exit_probe 12 1 3
t bin/exit_probe 12 1 0
