#!/bin/bash
# TEST INFRASTRUCTURE (debugging aid): instruments a line range of a device source and of an oracle source with TRACE lines (autotrace.py), rebuilds the
# CPU emulation and the oracle with tests/emu/trace.h, runs pass_diff at one pixel and prints the first intermediates whose bits differ.
# usage: trace_run.sh DENOISER W H FRAMES X Y DEVFILE:START:END:ATLINE ORAFILE:START:END:ATLINE [extra pass_diff options]
#   ATLINE = the line after which TRACE_AT(px, py) is inserted (px, py must be in scope there); the sources are restored afterwards.
cd "$(dirname "$0")/../.."
DEN=$1; W=$2; H=$3; FR=$4; X=$5; Y=$6; IFS=: read DF DS DE DA <<< "$7"; IFS=: read OF OS OE OA <<< "$8"; shift 8
export NRD_EMU_EXTRA_FLAGS="-include $PWD/tests/emu/trace.h -DNRD_TRACE_SIDE=\"dev\""
cp $DF /tmp/trace_dev.bak; cp $OF /tmp/trace_ora.bak
python tests/emu/autotrace.py instrument $DF $DS $DE; rm -f $DF.untraced
python tests/emu/autotrace.py instrument $OF $OS $OE; rm -f $OF.untraced
# TRACE_AT after the given ORIGINAL line: the instrumented file has extra lines, so find the original line's text
python - "$DF" /tmp/trace_dev.bak "$DA" "$OF" /tmp/trace_ora.bak "$OA" <<'PY'
import sys
for f, bak, at in ((sys.argv[1], sys.argv[2], int(sys.argv[3])), (sys.argv[4], sys.argv[5], int(sys.argv[6]))):
    anchor = open(bak).read().split("\n")[at - 1]
    lines = open(f).read().split("\n")
    k = lines.index(anchor, at - 1)  # the instrumented file only gained lines
    lines.insert(k + 1, "TRACE_AT(px, py);")
    open(f, "w").write("\n".join(lines))
PY
python tests/emu/build_emu.py 2>&1 | grep -E "error" -A5 | head -30
make -C oracle -s -j8 EXTRA="-include $PWD/tests/emu/trace.h -DNRD_TRACE_SIDE=\\\"ora\\\"" 2>&1 | grep -E "error" -A5 | head -30
NRD_TRACE_X=$X NRD_TRACE_Y=$Y OMP_NUM_THREADS=1 python tests/emu/pass_diff.py $DEN $W $H $FR "$@" 2>/tmp/trace.log >/dev/null
python tests/emu/autotrace.py compare /tmp/trace.log | head -${TRACE_HEAD:-25}
cp /tmp/trace_dev.bak $DF; cp /tmp/trace_ora.bak $OF
