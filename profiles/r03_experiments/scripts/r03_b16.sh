# round 3, batch 16: the samplers' sin / cos — own fp64 routine for [0, 2 pi] (csrc/pt_sincos.h) vs OCML's general double sincos (-DPT_OCML_SINCOS=1)
cd $GRAFT_REPO_ROOT
bash tools/ab.sh r03_b16 main ocml
