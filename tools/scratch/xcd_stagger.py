"""Negative result, round 4 (kept as a record; the `xcd_stagger` knob it drove was removed from the kernel again):
first-round workgroups of XCD x slept x * d * 64 cycles so that the XCDs' epilogue store bursts stop coinciding.
One MI355X, random operands, min of 3 x 10 launches, d = 0 / small / medium / one tile time spread over the 8 XCDs:
  w12 split3   65536 x 8192 x 3072 bf16 : 2307.4  2332.2  2335.7  2367.3 us
  in_proj f16   2048 x 65536 x 512      :  170.1   169.5   167.0   172.7 us
  in_proj split3 2048 x 65536 x 1536    :  319.7   325.3   333.7   343.4 us
  qkv f16      65536 x 1536 x 512       :  129.8   130.0   133.6   139.0 us
  w12 f16      65536 x 8192 x 1024      : 1037.4  1070.0  1073.1  1066.8 us
The same with the CUs of one XCD staggered against each other (start of round 4): 2.68 -> 2.68 / 2.81 ms (it also breaks the lock-step in
which the 32 workgroups of an XCD share their operand panels through its L2)."""
