#!/bin/bash
# usage: tools/viz_loop.sh <file.s> <kernel-name-substring>: prints the main loop of a kernel as one letter per instruction
# (M mfma, e v_exp, v other VALU, d ds_read, w s_waitcnt, B barrier, G LDS-DMA, X scratch, n s_nop, s other SALU, | branch)
S=$1; K=$2
START=$(grep -n "^_Z[A-Za-z0-9_]*$K[A-Za-z0-9_]*:" $S | head -1 | cut -d: -f1)
tail -n +$START $S | awk '{print} /s_endpgm/ {exit}' > /tmp/_k.s
L0=$(grep -n "Loop Header" /tmp/_k.s | head -1 | cut -d: -f1); LB=$(sed -n "${L0}p" /tmp/_k.s | cut -d: -f1)
L1=$(grep -n "s_branch $LB\|s_cbranch_scc1 $LB\|s_cbranch_scc0 $LB\|s_cbranch_vccnz $LB" /tmp/_k.s | tail -1 | cut -d: -f1)
awk -v a=$L0 -v b=$L1 'NR>=a && NR<=b' /tmp/_k.s > /tmp/_loop.s
echo "kernel at line $START; loop lines $L0..$L1 ($LB); scratch ops in loop: $(grep -c scratch_ /tmp/_loop.s); in kernel: $(grep -c scratch_ /tmp/_k.s)"
awk '{ if ($1 ~ /^v_mfma/) printf "M"; else if ($1 ~ /^v_exp/) printf "e"; else if ($1 ~ /^v_/) printf "v"; else if ($1 ~ /^ds_read/) printf "d"; else if ($1 ~ /^s_waitcnt/) printf "w"; else if ($1 ~ /^s_barrier/) printf "B\n"; else if ($1 ~ /^global_load_lds/) printf "G"; else if ($1 ~ /^s_cbranch|^s_branch/) printf "|"; else if ($1 ~ /^scratch/) printf "X"; else if ($1 ~ /^s_nop/) printf "n"; else if ($1 ~ /^s_/) printf "s"; else if ($1 ~ /^\.LBB/) printf "\n%s ", $1; }' /tmp/_loop.s; echo
