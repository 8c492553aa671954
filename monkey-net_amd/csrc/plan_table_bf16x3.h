// Launch plans of the forward / data-gradient GEMMs measured WITH gemm_bf16x3 = 1 (tools/plan_tune.py under MNK_TUNING=gemm_bf16x3=1
// on the MI355X, round 6: moving-gif @64^2 batch 32; cold launches incl. the split-K reduction and the statistics pass a split
// costs a norm layer).  The six-MFMA bf16 form shifts the balance towards wider tiles (its K loop is issue-bound: a 128-wide tile
// halves the loader / split work per MFMA): 25 rows, summed gain 84 us against make_plan's rule.  Consulted in front of
// plan_table.h when the tuning value is set.  {M, Cout, chunks, taps, phases, bm, bn, splits}
    {16384, 64, 8, 9, 1, 64, 64, 3},   // kp.enc1 dgrad: 41.5 -> 37.5 us
    {4096, 256, 8, 9, 1, 64, 64, 3},   // kp.enc2 fwd: 38.9 -> 37.2 us
    {4096, 128, 16, 9, 1, 64, 64, 4},   // kp.enc2 dgrad: 36.7 -> 35.4 us
    {256, 512, 64, 9, 1, 64, 128, 32},   // kp.enc4 dgrad: 35.5 -> 33.8 us
    {256, 1024, 16, 16, 1, 64, 128, 16},   // kp.dec1 dgrad: 33.1 -> 32.0 us
    {4096, 64, 16, 4, 4, 64, 64, 4},   // kp.dec3 fwd: 36.9 -> 35.2 us
    {4096, 256, 4, 16, 1, 64, 64, 3},   // kp.dec3 dgrad: 36.4 -> 34.3 us
    {65536, 35, 1, 9, 1, 64, 64, 1},   // kp.last dgrad: 19.6 -> 18.7 us
    {32768, 128, 4, 9, 1, 64, 128, 1},   // gen.app1 fwd: 46.0 -> 43.7 us
    {2048, 512, 16, 9, 1, 128, 128, 8},   // gen.app3 fwd: 53.6 -> 49.0 us
    {2048, 256, 32, 9, 1, 128, 128, 16},   // gen.app3 dgrad: 51.3 -> 48.1 us
    {512, 512, 64, 9, 1, 128, 128, 32},   // gen.app4 dgrad: 51.8 -> 48.0 us
    {128, 1024, 64, 9, 1, 64, 128, 32},   // gen.app5 dgrad: 37.0 -> 35.4 us
    {32768, 66, 4, 9, 1, 64, 128, 1},   // gen.dm.enc0 dgrad: 45.5 -> 40.4 us
    {2048, 128, 16, 9, 1, 64, 64, 8},   // gen.dm.enc2 dgrad: 27.7 -> 25.8 us
    {512, 256, 32, 9, 1, 64, 128, 32},   // gen.dm.enc3 dgrad: 28.5 -> 25.5 us
    {128, 512, 64, 9, 1, 64, 128, 32},   // gen.dm.enc4 dgrad: 30.8 -> 28.1 us
    {32, 1024, 32, 16, 1, 64, 64, 47},   // gen.dm.dec0 dgrad: 28.4 -> 27.4 us
    {32768, 13, 7, 9, 1, 128, 16, 4},   // gen.dm.last fwd: 40.4 -> 30.9 us
    {32, 1034, 64, 16, 1, 64, 64, 32},   // gen.dec.dec0 dgrad: 46.6 -> 45.0 us
    {128, 2058, 32, 16, 1, 64, 128, 12},   // gen.dec.dec1 dgrad: 65.9 -> 53.5 us
    {512, 256, 65, 4, 4, 128, 64, 8},   // gen.dec.dec2 fwd: 49.7 -> 46.4 us
    {512, 1034, 16, 16, 1, 64, 128, 6},   // gen.dec.dec2 dgrad: 58.1 -> 52.2 us
    {2048, 128, 33, 4, 4, 128, 64, 4},   // gen.dec.dec3 fwd: 49.6 -> 47.7 us
    {2048, 522, 8, 16, 1, 128, 64, 3},   // gen.dec.dec3 dgrad: 58.7 -> 53.3 us
