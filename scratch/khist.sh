#!/bin/bash
# usage: scratch/khist.sh <mangled-name-substring> : per-barrier-segment instruction histograms of a kernel in ${ASM:-/tmp/spmm_lds.s}
name=$1
awk -v n="$name" '$0 ~ "^_ZN.*"n".*:" {on=1} on{print} on && /s_endpgm/{exit}' ${ASM:-/tmp/spmm_lds.s} > /tmp/k.s
grep -n "s_barrier" /tmp/k.s | cut -d: -f1 | tr '\n' ' '; echo
prev=1
for b in $(grep -n "s_barrier" /tmp/k.s | cut -d: -f1) $(wc -l < /tmp/k.s); do
  echo "---- lines $prev-$b"
  sed -n "${prev},${b}p" /tmp/k.s | grep -v "^\s*;" | grep -v "^\." | awk '{print $1}' | sort | uniq -c | sort -rn | head -${TOP:-18} | paste - - - - - - | sed 's/  */ /g'
  prev=$b
done
