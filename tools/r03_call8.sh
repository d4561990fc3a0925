#!/bin/bash
cd /root/repo
timeout 600 python tools/k3_time.py v1 v2 v1 v2
