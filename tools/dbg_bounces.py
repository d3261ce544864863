import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd import Renderer, _ffi
scene = bench.build_scene(sys.argv[1] if len(sys.argv) > 1 else "dragon", 4)
r = Renderer(scene, options={_ffi.OPT_SEMANTICS:1,_ffi.OPT_SEEDING:1})
r.set_option(_ffi.OPT_COUNT_TRAVERSAL,1)
r.render_range(0,4); r.synchronize()
r.close()
