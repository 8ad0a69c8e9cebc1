# one-off option sweep on the write-through build (B = 32 fp32): have the tuned choices moved?
cd $GRAFT_REPO_ROOT
OPTS='[[("f4_share3",80),("f4_share2",20)],[("f4_share3",90),("f4_share2",10)],[("f4_share3",70),("f4_share2",15)],[("f4_share3",60),("f4_share2",25)],[("f4_share3",100),("f4_share2",0)],
[("s4",6)],[("s4",5)],[("s4",4)],[("s4",7)],
[("tps:1",13)],[("tps:1",20)],[("tps:1",25)],[("tps:1",16)],
[("tps:2",6)],[("tps:2",11)],[("tps:2",14)],[("tps:2",8)],
[("tps:3",7)],[("tps:3",10)],[("tps:3",18)],[("tps:3",13)],
[("nw:1",8)],[("nw:5",8)],[("nw:3",8)],
[("bwd_order",1)],[("bwd_order",2)],[("bwd_order",3)],
[("hoist",1)],[("xcd_map",1)],[("r3_xcd",3)],[("r3_xcd",0)],[("r3_xcd",2)]]' N=4000 timeout 300 python tools/ab_options.py 2>&1 | tail -40
