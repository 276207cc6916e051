#version 130
// REPO-AUTHORED STAND-IN, not the reference's file.  The reference's include/shaders/urdf_filter.vert cannot travel
// to the GPU box (reference sources are never copied into this repository), so bench.py's llvmpipe leg there runs this
// re-statement of the same two operations: clip position = MVP * vertex (include/shaders/urdf_filter.vert:4-5) and the
// camera-facing normal for the debug attachment (:7-8).  tests/test_oracle_vs_llvmpipe.py (development container)
// checks that Mesa produces bit-identical frames with this pair and with the reference's own pair.
out vec4 shaded_normal;

void main()
{
  vec3 n = gl_NormalMatrix * gl_Normal;
  shaded_normal = vec4(-n.x, n.y, -n.z, 1.0);
  gl_Position = gl_ModelViewProjectionMatrix * gl_Vertex;
}
