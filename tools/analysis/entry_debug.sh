#!/bin/bash
# old (lib_old) against new (lib) float images of a few workloads; summary to gpurun_out/entry/debug.log
mkdir -p gpurun_out/entry
D=tools/analysis/entry_debug.py
run() {
	OCRT_LIB_DIR=$PWD/opencl_raytracer_amd/lib_old timeout -k 10 300 python3 $D "$@" /dev/null > /dev/null 2>&1 || true
}
pair() {
	local w=$1; shift
	echo "== $w $*"
	OCRT_LIB_DIR=$PWD/opencl_raytracer_amd/lib_old timeout -k 10 300 python3 $D $w /tmp/a.npy "$@" || return 1
	timeout -k 10 300 python3 $D $w /tmp/b.npy "$@" || return 1
	python3 $D compare /tmp/a.npy /tmp/b.npy
}
{
	pair bunny_1080p_s64
} > gpurun_out/entry/debug.log 2>&1
tail -120 gpurun_out/entry/debug.log
