export FVH_COMMIT=49d041f7bc9c
bash tools/r05_artifacts.sh all 2>&1 | tail -60
