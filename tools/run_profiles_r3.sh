# Round-3 evidence run (ON the GPU box): kernel-trace stats of the headline bench at two step counts (per-step launch counts =
# difference / 8), the three PMC passes (FETCH_SIZE | WRITE_SIZE | MFMA busy + GUI active, never mixed with API traces), and
# kernel stats of the configs[3] / configs[4] modes.  Everything lands in gpurun_out/r3prof/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MI355X_GRAPHS=0   # live launches (what the auto trial keeps on this stack): no recording / trial steps inside the traces
O=$GRAFT_REPO_ROOT/gpurun_out/r3prof
mkdir -p $O
python -c "import bench; print(bench._source_hash())" > $O/source_hash.txt 2>/dev/null
finddb() { find $1 -name "*.db" | head -1; }
stats() {  # name, bench args...
  n=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/$n -o out -- python $GRAFT_REPO_ROOT/bench.py "$@" > $O/$n.json 2> $O/$n.err)
  db=$(finddb $O/$n)
  [ -n "$db" ] && python tools/rocpd_stats.py $db $O/$n > /dev/null
  rm -rf $O/$n
  tail -1 $O/$n.json | cut -c1-200
}
pmc() {  # name, counters
  n=$1; shift
  (cd /tmp && MI355X_WGRAD_STREAM=0 timeout 500 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/$n.json 2> $O/$n.err)
  db=$(finddb $O/$n)
  [ -n "$db" ] && python tools/pmc_dump.py $db $O/$n.pmc.json
  rm -rf $O/$n
}
PARTS=${R3_PARTS:-"stats pmc models"}   # e.g. R3_PARTS=stats: only the two kernel-trace runs of the headline bench
case " $PARTS " in *" stats "*)
stats stats_s4 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline
stats stats_s12 --steps 12 --warmup 2 --no-cpu-baseline --no-roofline
;; esac
case " $PARTS " in *" pmc "*)
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
pmc pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
;; esac
case " $PARTS " in *" models "*)
stats stats_transducer --model transducer --steps 3 --warmup 2 --no-roofline
stats stats_squeezeformer --model squeezeformer --size medium --steps 3 --warmup 2 --no-roofline
;; esac
ls -la $O
