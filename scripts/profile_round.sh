TAG=r06
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
PROFILE_STEPS=20 PROFILE_WARMUP=5 bash scripts/profile_gpu.sh ${TAG} --no-extras > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_pair --no-extras --loop pair --pipeline 0 > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_space_invaders --game space_invaders --no-extras > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_amidar --game amidar --no-extras > /dev/null 2>&1
bash scripts/profile_gpu.sh ${TAG}_breakout_4096 --envs 4096 --no-extras > /dev/null 2>&1
for g in breakout space_invaders amidar; do bash scripts/profile_gpu.sh ${TAG}_${g}_4096_pair --game $g --envs 4096 --no-extras --loop pair --pipeline 0 > /dev/null 2>&1; done
bash scripts/profile_gpu.sh ${TAG}_breakout_8192_gather --envs 8192 --with-gather --no-extras > /dev/null 2>&1
find gpurun_out -size +8M -delete; du -sh gpurun_out/prof_${TAG}* 2>/dev/null | tail -12
cat gpurun_out/prof_r06/csrc_sha16.txt
