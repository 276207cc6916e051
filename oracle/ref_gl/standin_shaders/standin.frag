#version 140
// REPO-AUTHORED STAND-IN, not the reference's file (see standin.vert).  Same uniform interface as the reference's
// fragment program (the harness sets them by name) and the same per-fragment arithmetic as
// include/shaders/urdf_filter.frag:14-35: sensor depth fetched by pixel index, window z turned into metres with
// (n*f/(n-f)) / (z - f/(f-n)), "filter" = sensor > virtual - max_diff, outputs: sensor, masked depth, normal, mask.
in vec4 shaded_normal;
uniform int width;
uniform int height;
uniform samplerBuffer depth_texture;
uniform float replace_value;
uniform float z_near;
uniform float z_far;
uniform float max_diff;

void main(void)
{
  int texel = int(gl_FragCoord.y) * width + int(gl_FragCoord.x);
  float measured = texelFetch(depth_texture, texel).x;
  float rendered = (z_near * z_far / (z_near - z_far)) / (gl_FragCoord.z - z_far / (z_far - z_near));
  float hit = float(measured > (rendered - max_diff));
  vec4 grey = vec4(measured, measured, measured, 1.0);
  gl_FragData[0] = grey;
  gl_FragData[1] = mix(grey, vec4(replace_value, 0.0, 0.0, 1.0), hit);
  gl_FragData[2] = shaded_normal * 0.5 + 0.5;
  gl_FragData[3] = vec4(hit, hit, hit, 0.0);
}
