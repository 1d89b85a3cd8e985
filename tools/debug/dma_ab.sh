export PIPE_DEPTH=4 PIPE_STEPS=120
run() { echo "== $1"; shift; env "$@" timeout 300 python tools/pipeline_probe.py 2>&1 | grep -v amdgpu.ids; }
run baseline X=1
run "map D:H,B:I" PIPE_TILE_MAP=D:H,B:I
run "map D:H,B:I,C:J" PIPE_TILE_MAP=D:H,B:I,C:J
run "wino I/A" "PIPE_WINO_RULE='I' if Cout >= 256 else 'A'"
run "wino I/I" "PIPE_WINO_RULE='I'"
run "wino I(gates only: R<=8192)/B" "PIPE_WINO_RULE='I' if (Cout >= 256 and R <= 8192) else 'B'"
run "wino I/A + map" "PIPE_WINO_RULE='I' if Cout >= 256 else 'A'" PIPE_TILE_MAP=D:H,B:I
run "wino I/I + map" "PIPE_WINO_RULE='I'" PIPE_TILE_MAP=D:H,B:I
run baseline X=1
