#!/bin/bash
# Copy the condensed artefacts of `tools/run_profiles.sh <tag>` from gpurun_out/ (scratch) into profiles/ (tracked).
# Run in the build container after the gpurun calls have merged their output back.
set -e
tag=${1:-r03}
out=gpurun_out
for m in learn inference learn_k100 inference_k100 transfer; do cp $out/${tag}_${m}_bench_stats.md profiles/${tag}_${m}_bench_stats.md; done
cp $out/${tag}_hbm_traffic.json profiles/hbm_traffic.json
cp $out/${tag}_hbm_traffic_k100.json profiles/hbm_traffic_k100.json
for f in mfma_utilisation mfma_utilisation_k100; do     # the `mfma` phase (matrix-pipe utilisation; roofline.mfma_utilisation of the bench line)
  if [ -f $out/${tag}_$f.json ]; then cp $out/${tag}_$f.json profiles/$f.json; fi
done
for f in ${tag}_bench_line ${tag}_bench_line_inference ${tag}_bench_line_transfer; do grep "^{" $out/$f.json | tail -1 > profiles/$f.json; done
{
  echo "# One steady-state learning step of \`bench.py\` (rocprofv3 kernel trace, run ${tag}) — where the time goes"; echo
  echo "## headline step (cached pseudo-labels: 1 forward + 1 backward of the frozen classifier)"; echo; echo '```'
  cat $out/${tag}_learn_step_breakdown.txt; echo '```'; echo
  echo "## the reference's op sequence (\`--cache-labels 0\`: labels recomputed, 2 forwards + 1 backward)"; echo; echo '```'
  cat $out/${tag}_learn_step_breakdown_recomputed_labels.txt; echo '```'
} > profiles/${tag}_learn_step_breakdown.md
{
  echo "# HBM traffic per launch from PMC counters — ${tag} (kernel source hash $(cat $out/${tag}_kernel_source_hash.txt))"; echo
  cat tools/pmc_traffic_preamble.md; echo; echo "## K = 50"; echo; echo '```'
  grep -v amdgpu.ids $out/${tag}_hbm_traffic.txt; echo '```'; echo; echo "## K = 100"; echo; echo '```'
  grep -v amdgpu.ids $out/${tag}_hbm_traffic_k100.txt; echo '```'
} > profiles/${tag}_hbm_traffic_pmc.md
echo "collected into profiles/ (kernel source hash $(cat $out/${tag}_kernel_source_hash.txt))"
