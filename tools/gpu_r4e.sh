#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/coop_profile.py ${1:-profile:1} 2>&1 | tail -6 | cut -c1-700
