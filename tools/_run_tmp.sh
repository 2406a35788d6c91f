for c in 2 4 8 64; do echo "== BPM_GRAPH=1 chunk $c"; BPM_GRAPH=1 BPM_GRAPH_CHUNK=$c python tools/window_anatomy.py 5 2>&1 | grep -v amdgpu | grep -A12 "^K "; done
echo "== stream launches"; python tools/window_anatomy.py 5 2>&1 | grep -v amdgpu | grep -A12 "^K "
