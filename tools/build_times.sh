#!/bin/bash
# CPU seconds (user + sys) of every compilation unit of the default library, compiled into a scratch directory (the tree's objects stay
# as they are): `tools/build_times.sh [DEV=1] > profiles/rNN_build_times.log`.  Eight units at a time; the figure is CPU time, not wall.
set -u
cd "$(dirname "$0")/../pyfft_amd/csrc"
OUT=$(mktemp -d /tmp/mifft_build_times.XXXXXX)
UNITS=$(make -pn OBJDIR=$OUT "$@" 2>/dev/null | sed -n 's/^OBJS *:\?= *//p' | head -1)
one() {
    o=$1
    TIMEFORMAT="%U %S"
    { time make -s OBJDIR=$(dirname $o) "${@:2}" $o >/dev/null 2>$o.err; } 2>$o.time
    awk -v u=$(basename $o .o) '{printf "%-28s %7.1f s\n", u, $1 + $2}' $o.time
}
export -f one
echo $UNITS | tr ' ' '\n' | grep -v '^$' | xargs -P 8 -I{} bash -c "one {} $*" | sort -k2 -n -r | tee $OUT/all.txt
awk '{t += $2} END {printf "%-28s %7.1f s = %.1f CPU-minutes\n", "TOTAL", t, t / 60}' $OUT/all.txt
rm -rf $OUT
