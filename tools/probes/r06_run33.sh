#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python examples/train_ppo_cnn_cueframes.py 2>&1 | grep -v amdgpu.ids | tail -9
