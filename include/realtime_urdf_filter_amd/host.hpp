// host.hpp -- C++ host-side support above the C ABI (include/rtuf.h), header-only, no dependencies.
//
// Stands in for what the reference obtains from third-party libraries that are not part of it:
//   tf / Bullet LinearMath   tf::Transform, tf::Quaternion, getOpenGLMatrix   -> rtuf_host::Transform
//   urdfdom                  urdf::Model::initString, links, visuals, joints   -> rtuf_host::UrdfModel
//   freeglut                 glutSolidCube / Sphere / Cylinder tessellation    -> rtuf_host::*_draws
//   Assimp                   mesh import (STL, Collada, OBJ)                   -> rtuf_host::load_mesh (load_stl / load_collada / load_obj)
//   TF tree                  lookupTransform                                   -> rtuf_host::TransformProvider,
//                                                                                 forward_kinematics
// Reference call sites: src/urdf_renderer.cpp:67-190, src/renderable.cpp:59-170, :306-452,
// src/urdf_filter.cpp:520-534, :602-614.  The arithmetic is restated from the libraries'
// published sources; the Python twin (realtime_urdf_filter_amd/urdf.py, geometry.py) is tested
// against the same golden fixtures.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <initializer_list>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

namespace rtuf_host {

// ---------------------------------------------------------------------------------------------
// tf-style transform algebra (doubles)
// ---------------------------------------------------------------------------------------------
struct Vec3 { double x = 0, y = 0, z = 0; };
struct Quat { double x = 0, y = 0, z = 0, w = 1; };

struct Transform {
  double m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};   // basis, row-major
  Vec3 o;                                              // origin

  static Transform from_quaternion(const Quat& q, const Vec3& origin = Vec3())
  {
    // Matrix3x3::setRotation
    Transform t;
    const double d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    const double s = 2.0 / d;
    const double xs = q.x * s, ys = q.y * s, zs = q.z * s;
    const double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    const double xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
    const double yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    t.m[0][0] = 1.0 - (yy + zz); t.m[0][1] = xy - wz; t.m[0][2] = xz + wy;
    t.m[1][0] = xy + wz; t.m[1][1] = 1.0 - (xx + zz); t.m[1][2] = yz - wx;
    t.m[2][0] = xz - wy; t.m[2][1] = yz + wx; t.m[2][2] = 1.0 - (xx + yy);
    t.o = origin;
    return t;
  }

  Vec3 apply(const Vec3& v) const
  {
    return {m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z + o.x, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z + o.y,
            m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z + o.z};
  }

  Transform operator*(const Transform& b) const
  {
    Transform r;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) r.m[i][j] = m[i][0] * b.m[0][j] + m[i][1] * b.m[1][j] + m[i][2] * b.m[2][j];
    r.o = apply(b.o);
    return r;
  }

  Transform inverse() const
  {
    Transform r;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) r.m[i][j] = m[j][i];
    const Vec3 n{-o.x, -o.y, -o.z};
    r.o = {r.m[0][0] * n.x + r.m[0][1] * n.y + r.m[0][2] * n.z, r.m[1][0] * n.x + r.m[1][1] * n.y + r.m[1][2] * n.z,
           r.m[2][0] * n.x + r.m[2][1] * n.y + r.m[2][2] * n.z};
    return r;
  }

  Quat rotation() const   // Matrix3x3::getRotation
  {
    const double trace = m[0][0] + m[1][1] + m[2][2];
    double t[4];
    if (trace > 0.0) {
      double s = std::sqrt(trace + 1.0);
      t[3] = s * 0.5;
      s = 0.5 / s;
      t[0] = (m[2][1] - m[1][2]) * s;
      t[1] = (m[0][2] - m[2][0]) * s;
      t[2] = (m[1][0] - m[0][1]) * s;
    } else {
      const int i = m[0][0] < m[1][1] ? (m[1][1] < m[2][2] ? 2 : 1) : (m[0][0] < m[2][2] ? 2 : 0);
      const int j = (i + 1) % 3, k = (i + 2) % 3;
      double s = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
      t[i] = s * 0.5;
      s = 0.5 / s;
      t[3] = (m[k][j] - m[j][k]) * s;
      t[j] = (m[j][i] + m[i][j]) * s;
      t[k] = (m[k][i] + m[i][k]) * s;
    }
    return {t[0], t[1], t[2], t[3]};
  }

  void opengl_matrix(double g[16]) const   // getOpenGLMatrix: column-major
  {
    g[0] = m[0][0]; g[1] = m[1][0]; g[2] = m[2][0]; g[3] = 0;
    g[4] = m[0][1]; g[5] = m[1][1]; g[6] = m[2][1]; g[7] = 0;
    g[8] = m[0][2]; g[9] = m[1][2]; g[10] = m[2][2]; g[11] = 0;
    g[12] = o.x; g[13] = o.y; g[14] = o.z; g[15] = 1;
  }
};

inline Quat quaternion_from_rpy(double roll, double pitch, double yaw)   // urdf::Rotation::setFromRPY
{
  const double phi = roll / 2.0, the = pitch / 2.0, psi = yaw / 2.0;
  Quat q;
  q.x = std::sin(phi) * std::cos(the) * std::cos(psi) - std::cos(phi) * std::sin(the) * std::sin(psi);
  q.y = std::cos(phi) * std::sin(the) * std::cos(psi) + std::sin(phi) * std::cos(the) * std::sin(psi);
  q.z = std::cos(phi) * std::cos(the) * std::sin(psi) - std::sin(phi) * std::sin(the) * std::cos(psi);
  q.w = std::cos(phi) * std::cos(the) * std::cos(psi) + std::sin(phi) * std::sin(the) * std::sin(psi);
  const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  if (n > 0) { q.x /= n; q.y /= n; q.z /= n; q.w /= n; }
  return q;
}

inline Transform pose_to_transform(const Vec3& xyz, const Vec3& rpy)   // src/urdf_renderer.cpp:160-164
{
  Quat q = quaternion_from_rpy(rpy.x, rpy.y, rpy.z);
  const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x /= n; q.y /= n; q.z /= n; q.w /= n;
  return Transform::from_quaternion(q, xyz);
}

// tf::TransformListener stand-in.  lookup() returns false on failure (tf::TransformException).
struct TransformProvider {
  virtual ~TransformProvider() = default;
  // transform taking points from `source` into `target`
  virtual bool lookup(const std::string& target, const std::string& source, Transform& out) const = 0;
};

struct StaticTransformProvider : TransformProvider {
  std::map<std::string, Transform> frames;   // name -> (root <- frame)
  bool lookup(const std::string& target, const std::string& source, Transform& out) const override
  {
    auto t = frames.find(target), s = frames.find(source);
    if (t == frames.end() || s == frames.end()) return false;
    out = t->second.inverse() * s->second;
    return true;
  }
};

// ---------------------------------------------------------------------------------------------
// minimal XML + URDF
// ---------------------------------------------------------------------------------------------
struct XmlNode {
  std::string tag;
  std::map<std::string, std::string> attr;
  std::vector<XmlNode> children;
  std::string text;             // character data directly inside the element (Collada arrays), concatenated
  std::vector<const XmlNode*> all(const std::string& t) const
  {
    std::vector<const XmlNode*> out;
    for (const auto& c : children) if (c.tag == t) out.push_back(&c);
    return out;
  }
  const XmlNode* path(std::initializer_list<const char*> tags) const
  {
    const XmlNode* n = this;
    for (const char* t : tags) { n = n->child(t); if (!n) return nullptr; }
    return n;
  }
  const XmlNode* child(const std::string& t) const
  {
    for (const auto& c : children) if (c.tag == t) return &c;
    return nullptr;
  }
  std::string get(const std::string& k, const std::string& dflt = "") const
  {
    auto it = attr.find(k);
    return it == attr.end() ? dflt : it->second;
  }
};

class XmlParser {
 public:
  explicit XmlParser(const std::string& s) : s_(s) {}
  XmlNode parse_document()
  {
    skip_misc();
    XmlNode root;
    if (!parse_element(root)) throw std::runtime_error("XML: no root element");
    return root;
  }

 private:
  const std::string& s_;
  size_t p_ = 0;
  void skip_ws() { while (p_ < s_.size() && std::isspace((unsigned char)s_[p_])) p_++; }
  bool starts(const char* lit) const { return s_.compare(p_, std::strlen(lit), lit) == 0; }
  void skip_misc(std::string* text = nullptr)
  {
    for (;;) {
      // text (including stray '>' as in the reference's example URDF) up to the next '<'
      const size_t t0 = p_;
      while (p_ < s_.size() && s_[p_] != '<') p_++;
      if (text && p_ > t0) { text->append(s_, t0, p_ - t0); text->push_back(' '); }
      if (p_ >= s_.size()) return;
      if (starts("<!--")) { const size_t e = s_.find("-->", p_); p_ = e == std::string::npos ? s_.size() : e + 3; }
      else if (starts("<?")) { const size_t e = s_.find("?>", p_); p_ = e == std::string::npos ? s_.size() : e + 2; }
      else if (starts("<!")) { const size_t e = s_.find('>', p_); p_ = e == std::string::npos ? s_.size() : e + 1; }
      else return;
    }
  }
  static bool name_char(char c) { return std::isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.'; }
  bool parse_element(XmlNode& n)
  {
    if (p_ >= s_.size() || s_[p_] != '<' || starts("</")) return false;
    p_++;
    size_t b = p_;
    while (p_ < s_.size() && name_char(s_[p_])) p_++;
    n.tag = s_.substr(b, p_ - b);
    for (;;) {
      skip_ws();
      if (p_ >= s_.size()) throw std::runtime_error("XML: unterminated tag <" + n.tag);
      if (s_[p_] == '/') { p_ += 2; return true; }           // "/>"
      if (s_[p_] == '>') { p_++; break; }
      b = p_;
      while (p_ < s_.size() && name_char(s_[p_])) p_++;
      const std::string key = s_.substr(b, p_ - b);
      skip_ws();
      if (p_ >= s_.size() || s_[p_] != '=') throw std::runtime_error("XML: attribute without value in <" + n.tag);
      p_++;
      skip_ws();
      const char q = s_[p_++];
      b = p_;
      while (p_ < s_.size() && s_[p_] != q) p_++;
      n.attr[key] = s_.substr(b, p_ - b);
      p_++;
    }
    for (;;) {
      skip_misc(&n.text);
      if (p_ >= s_.size()) throw std::runtime_error("XML: missing </" + n.tag + ">");
      if (starts("</")) {
        const size_t e = s_.find('>', p_);
        p_ = e == std::string::npos ? s_.size() : e + 1;
        return true;
      }
      XmlNode c;
      if (parse_element(c)) n.children.push_back(std::move(c));
    }
  }
};

struct UrdfGeometry {
  enum Kind { BOX, CYLINDER, SPHERE, MESH } kind = BOX;
  Vec3 size;                 // box
  double radius = 0, length = 0;
  std::string filename;      // mesh
  Vec3 scale{1, 1, 1};
};
struct UrdfVisual { Vec3 xyz, rpy; UrdfGeometry geometry; };
struct UrdfLink { std::string name; std::vector<UrdfVisual> visual_array, collision_array; };
struct UrdfJoint {
  std::string name, type, parent, child;
  Vec3 xyz, rpy, axis{1, 0, 0};
  double lower = 0, upper = 0;
  // <mimic joint multiplier offset>: position = multiplier * position(mimic_joint) + offset (urdfdom defaults 1, 0)
  std::string mimic_joint;
  double mimic_multiplier = 1, mimic_offset = 0;
};

inline Vec3 parse_vec3(const std::string& s, const Vec3& dflt)
{
  if (s.empty()) return dflt;
  Vec3 v;
  if (std::sscanf(s.c_str(), "%lf %lf %lf", &v.x, &v.y, &v.z) != 3) throw std::runtime_error("URDF: expected 3 numbers in '" + s + "'");
  return v;
}

struct UrdfModel {
  std::string name;
  std::map<std::string, UrdfLink> links;     // std::map: iteration by name, like urdf::Model::getLinks
  std::map<std::string, UrdfJoint> joints;

  static UrdfModel from_string(const std::string& xml)
  {
    XmlParser parser(xml);
    const XmlNode root = parser.parse_document();
    if (root.tag != "robot") throw std::runtime_error("URDF root element must be <robot>");
    UrdfModel m;
    m.name = root.get("name");
    for (const XmlNode& le : root.children) {
      if (le.tag == "link") {
        UrdfLink link;
        link.name = le.get("name");
        for (const XmlNode& ve : le.children) {
          if (ve.tag != "visual" && ve.tag != "collision") continue;
          const XmlNode* g = ve.child("geometry");
          if (!g || g->children.empty()) continue;
          const XmlNode& c = g->children[0];
          UrdfVisual v;
          if (c.tag == "box") { v.geometry.kind = UrdfGeometry::BOX; v.geometry.size = parse_vec3(c.get("size"), Vec3()); }
          else if (c.tag == "cylinder") { v.geometry.kind = UrdfGeometry::CYLINDER; v.geometry.radius = std::stod(c.get("radius")); v.geometry.length = std::stod(c.get("length")); }
          else if (c.tag == "sphere") { v.geometry.kind = UrdfGeometry::SPHERE; v.geometry.radius = std::stod(c.get("radius")); }
          else if (c.tag == "mesh") { v.geometry.kind = UrdfGeometry::MESH; v.geometry.filename = c.get("filename"); v.geometry.scale = parse_vec3(c.get("scale"), Vec3{1, 1, 1}); }
          else continue;
          if (const XmlNode* o = ve.child("origin")) { v.xyz = parse_vec3(o->get("xyz"), Vec3()); v.rpy = parse_vec3(o->get("rpy"), Vec3()); }
          (ve.tag == "visual" ? link.visual_array : link.collision_array).push_back(v);
        }
        m.links[link.name] = link;
      } else if (le.tag == "joint") {
        UrdfJoint j;
        j.name = le.get("name");
        j.type = le.get("type");
        if (const XmlNode* o = le.child("origin")) { j.xyz = parse_vec3(o->get("xyz"), Vec3()); j.rpy = parse_vec3(o->get("rpy"), Vec3()); }
        if (const XmlNode* a = le.child("axis")) j.axis = parse_vec3(a->get("xyz"), Vec3{1, 0, 0});
        if (const XmlNode* l = le.child("limit")) { j.lower = std::stod(l->get("lower", "0")); j.upper = std::stod(l->get("upper", "0")); }
        if (const XmlNode* mm = le.child("mimic")) {
          j.mimic_joint = mm->get("joint");
          j.mimic_multiplier = std::stod(mm->get("multiplier", "1"));
          j.mimic_offset = std::stod(mm->get("offset", "0"));
        }
        const XmlNode* p = le.child("parent");
        const XmlNode* c = le.child("child");
        if (!p || !c) throw std::runtime_error("URDF joint " + j.name + " lacks parent/child");
        j.parent = p->get("link");
        j.child = c->get("link");
        m.joints[j.name] = j;
      }
    }
    return m;
  }

  std::string root_link() const
  {
    std::unordered_set<std::string> children;
    for (const auto& j : joints) children.insert(j.second.child);
    std::string root;
    int n = 0;
    for (const auto& l : links) if (!children.count(l.first)) { root = l.first; n++; }
    if (n != 1) throw std::runtime_error("URDF must have exactly one root link");
    return root;
  }
};

inline Transform joint_motion(const UrdfJoint& j, double q)
{
  if (j.type == "revolute" || j.type == "continuous") {
    double n = std::sqrt(j.axis.x * j.axis.x + j.axis.y * j.axis.y + j.axis.z * j.axis.z);
    Vec3 a = n > 0 ? Vec3{j.axis.x / n, j.axis.y / n, j.axis.z / n} : Vec3{1, 0, 0};
    const double s = std::sin(0.5 * q);
    return Transform::from_quaternion({a.x * s, a.y * s, a.z * s, std::cos(0.5 * q)});
  }
  if (j.type == "prismatic") {
    Transform t;
    t.o = {j.axis.x * q, j.axis.y * q, j.axis.z * q};
    return t;
  }
  return Transform();
}

// Joint positions with every <mimic> joint filled in from the joint it follows (chains followed, explicit positions
// kept, a cycle throws): what joint_state_publisher does before robot_state_publisher makes the TF frames the reference
// looks up (src/urdf_renderer.cpp:173-190).  The PR2's gripper fingers are mimic joints.
inline std::map<std::string, double> resolve_mimic(const UrdfModel& m, std::map<std::string, double> q)
{
  for (const auto& kv : m.joints) {
    if (kv.second.mimic_joint.empty() || q.count(kv.first)) continue;
    // walk the chain up to a joint with a known position (or no mimic), then fill it back down
    std::vector<const UrdfJoint*> chain;
    const UrdfJoint* j = &kv.second;
    while (j && !j->mimic_joint.empty() && !q.count(j->name)) {
      if (chain.size() > m.joints.size()) throw std::runtime_error("URDF: mimic cycle through joint " + kv.first);
      chain.push_back(j);
      auto it = m.joints.find(j->mimic_joint);
      j = it == m.joints.end() ? nullptr : &it->second;
    }
    double v = 0.0;
    if (j) { auto it = q.find(j->name); v = it == q.end() ? 0.0 : it->second; }
    for (size_t i = chain.size(); i-- > 0;) {
      v = chain[i]->mimic_multiplier * v + chain[i]->mimic_offset;
      q[chain[i]->name] = v;
    }
  }
  return q;
}

// root <- link for every link (replaces the per-link TF lookups of update_link_transforms)
inline std::map<std::string, Transform> forward_kinematics(const UrdfModel& m, const std::map<std::string, double>& q_in = {})
{
  const std::map<std::string, double> q = resolve_mimic(m, q_in);
  std::map<std::string, std::vector<const UrdfJoint*>> by_parent;
  for (const auto& j : m.joints) by_parent[j.second.parent].push_back(&j.second);
  std::map<std::string, Transform> out;
  std::vector<std::string> stack{m.root_link()};
  out[stack[0]] = Transform();
  while (!stack.empty()) {
    const std::string p = stack.back();
    stack.pop_back();
    for (const UrdfJoint* j : by_parent[p]) {
      auto it = q.find(j->name);
      out[j->child] = out[p] * pose_to_transform(j->xyz, j->rpy) * joint_motion(*j, it == q.end() ? 0.0 : it->second);
      stack.push_back(j->child);
    }
  }
  return out;
}

// ---------------------------------------------------------------------------------------------
// geometry: GL draw calls as indexed triangles in the driver's assembly order
// ---------------------------------------------------------------------------------------------
struct DrawCall {
  int pre_op = 0;            // RTUF_OP_*
  float op[3] = {0, 0, 0};
  std::vector<float> verts;  // xyz
  std::vector<uint32_t> tris;
};

enum { GL_TRIANGLE_FAN_ = 6, GL_QUADS_ = 7, GL_QUAD_STRIP_ = 8 };
struct GlPrim { int mode; std::vector<double> verts; };

inline DrawCall prims_to_draw(const std::vector<GlPrim>& prims, int pre_op = 0, const float* op = nullptr)
{
  DrawCall d;
  d.pre_op = pre_op;
  if (op) { d.op[0] = op[0]; d.op[1] = op[1]; d.op[2] = op[2]; }
  for (const GlPrim& p : prims) {
    const uint32_t base = (uint32_t)(d.verts.size() / 3);
    for (double v : p.verts) d.verts.push_back((float)v);          // glVertex3d -> float
    const uint32_t n = (uint32_t)(p.verts.size() / 3);
    auto tri = [&](uint32_t a, uint32_t b, uint32_t c) { d.tris.push_back(base + a); d.tris.push_back(base + b); d.tris.push_back(base + c); };
    if (p.mode == GL_TRIANGLE_FAN_) for (uint32_t i = 2; i < n; i++) tri(0, i - 1, i);
    else if (p.mode == GL_QUAD_STRIP_) for (uint32_t i = 3; i < n; i += 2) { tri(i - 3, i - 2, i); tri(i - 1, i - 3, i); }
    else if (p.mode == GL_QUADS_) for (uint32_t i = 3; i < n; i += 4) { tri(i - 3, i - 2, i); tri(i - 2, i - 1, i); }
    else throw std::runtime_error("unsupported GL primitive mode");
  }
  return d;
}

inline void circle_table(int n, std::vector<double>& sint, std::vector<double>& cost)   // fghCircleTable
{
  const int size = std::abs(n);
  const double angle = 2 * M_PI / (double)(n == 0 ? 1 : n);
  sint.assign(size + 1, 0.0);
  cost.assign(size + 1, 0.0);
  sint[0] = 0.0; cost[0] = 1.0;
  for (int i = 1; i < size; i++) { sint[i] = std::sin(angle * i); cost[i] = std::cos(angle * i); }
  sint[size] = sint[0]; cost[size] = cost[0];
}

inline std::vector<GlPrim> cube_prims(float size_f)   // glutSolidCube
{
  const double s = (double)size_f * 0.5, P = s, N = -s;
  GlPrim q{GL_QUADS_, {P, N, P, P, N, N, P, P, N, P, P, P, P, P, P, P, P, N, N, P, N, N, P, P, P, P, P, N, P, P, N, N, P, P, N, P,
                       N, N, P, N, P, P, N, P, N, N, N, N, N, N, P, N, N, N, P, N, N, P, N, P, N, N, N, N, P, N, P, P, N, P, N, N}};
  return {q};
}

// RenderableBox::render (src/renderable.cpp:107-170): the VBO box, then the scaled glutSolidCube (quirk Q1)
inline std::vector<DrawCall> box_draws(float dx, float dy, float dz)
{
  const float X = 0.5f * dx, Y = 0.5f * dy, Z = 0.5f * dz;
  const float v[24][3] = {{X, Y, -Z}, {-X, Y, -Z}, {-X, Y, Z}, {X, Y, Z}, {X, -Y, Z}, {-X, -Y, Z}, {-X, -Y, -Z}, {X, -Y, -Z},
                          {X, Y, Z}, {-X, Y, Z}, {-X, -Y, Z}, {X, -Y, Z}, {X, -Y, -Z}, {-X, -Y, -Z}, {-X, Y, -Z}, {X, Y, -Z},
                          {-X, Y, Z}, {-X, Y, -Z}, {-X, -Y, -Z}, {-X, -Y, Z}, {X, Y, -Z}, {X, Y, Z}, {X, -Y, Z}, {X, -Y, -Z}};
  DrawCall first;
  for (auto& p : v) { first.verts.push_back(p[0]); first.verts.push_back(p[1]); first.verts.push_back(p[2]); }
  for (uint32_t i = 0; i < 24; i += 4) { for (uint32_t k : {i, i + 1, i + 3, i + 1, i + 2, i + 3}) first.tris.push_back(k); }
  const float op[3] = {dx, dy, dz};
  return {first, prims_to_draw(cube_prims(dx), 1, op)};
}

inline std::vector<DrawCall> sphere_draws(float radius_f, int slices = 10, int stacks = 10)   // glutSolidSphere
{
  const double radius = radius_f;
  std::vector<double> sint1, cost1, sint2, cost2;
  circle_table(-slices, sint1, cost1);
  circle_table(stacks * 2, sint2, cost2);
  std::vector<GlPrim> prims;
  double z1 = cost2[stacks > 0 ? 1 : 0], r1 = sint2[stacks > 0 ? 1 : 0], z0, r0;
  GlPrim top{GL_TRIANGLE_FAN_, {0.0, 0.0, radius}};
  for (int j = slices; j >= 0; j--) { top.verts.push_back(cost1[j] * r1 * radius); top.verts.push_back(sint1[j] * r1 * radius); top.verts.push_back(z1 * radius); }
  prims.push_back(top);
  for (int i = 1; i < stacks - 1; i++) {
    z0 = z1; z1 = cost2[i + 1];
    r0 = r1; r1 = sint2[i + 1];
    GlPrim strip{GL_QUAD_STRIP_, {}};
    for (int j = 0; j <= slices; j++) {
      strip.verts.push_back(cost1[j] * r1 * radius); strip.verts.push_back(sint1[j] * r1 * radius); strip.verts.push_back(z1 * radius);
      strip.verts.push_back(cost1[j] * r0 * radius); strip.verts.push_back(sint1[j] * r0 * radius); strip.verts.push_back(z0 * radius);
    }
    prims.push_back(strip);
  }
  z0 = z1; r0 = r1;
  GlPrim bot{GL_TRIANGLE_FAN_, {0.0, 0.0, -radius}};
  for (int j = 0; j <= slices; j++) { bot.verts.push_back(cost1[j] * r0 * radius); bot.verts.push_back(sint1[j] * r0 * radius); bot.verts.push_back(z0 * radius); }
  prims.push_back(bot);
  return {prims_to_draw(prims)};
}

// glTranslatef(0,0,-length/2); glutSolidCylinder(radius, length, 10, 10) (src/renderable.cpp:92-98)
inline std::vector<DrawCall> cylinder_draws(float radius_f, float length_f, int slices = 10, int stacks = 10)
{
  const double radius = radius_f, height = length_f;
  std::vector<double> sint, cost;
  circle_table(-slices, sint, cost);
  const double zstep = height / (stacks > 0 ? stacks : 1);
  std::vector<GlPrim> prims;
  GlPrim base{GL_TRIANGLE_FAN_, {0.0, 0.0, 0.0}};
  for (int j = 0; j <= slices; j++) { base.verts.push_back(cost[j] * radius); base.verts.push_back(sint[j] * radius); base.verts.push_back(0.0); }
  prims.push_back(base);
  GlPrim top{GL_TRIANGLE_FAN_, {0.0, 0.0, height}};
  for (int j = slices; j >= 0; j--) { top.verts.push_back(cost[j] * radius); top.verts.push_back(sint[j] * radius); top.verts.push_back(height); }
  prims.push_back(top);
  double z0 = 0.0, z1 = zstep;
  for (int i = 1; i <= stacks; i++) {
    if (i == stacks) z1 = height;
    GlPrim strip{GL_QUAD_STRIP_, {}};
    for (int j = 0; j <= slices; j++) {
      strip.verts.push_back(cost[j] * radius); strip.verts.push_back(sint[j] * radius); strip.verts.push_back(z0);
      strip.verts.push_back(cost[j] * radius); strip.verts.push_back(sint[j] * radius); strip.verts.push_back(z1);
    }
    prims.push_back(strip);
    z0 = z1; z1 += zstep;
  }
  const float op[3] = {0.0f, 0.0f, -length_f / 2};
  return {prims_to_draw(prims, 2, op)};
}

// RenderableMesh::render: glScalef(scale) + indexed GL_TRIANGLES
inline std::vector<DrawCall> mesh_draws(const std::vector<float>& verts, const std::vector<uint32_t>& tris, float sx, float sy, float sz)
{
  DrawCall d;
  d.pre_op = 1;
  d.op[0] = sx; d.op[1] = sy; d.op[2] = sz;
  d.verts = verts;
  d.tris = tris;
  return {d};
}

// Binary (incl. headers starting with "solid", README.md:121-143) and ASCII STL; no vertex welding.
inline bool load_stl(const std::string& data, std::vector<float>& verts, std::vector<uint32_t>& tris)
{
  verts.clear();
  tris.clear();
  if (data.size() >= 84) {
    uint32_t n;
    std::memcpy(&n, data.data() + 80, 4);
    if (84 + 50ull * n == data.size()) {
      for (uint32_t i = 0; i < n; i++) {
        float v[9];
        std::memcpy(v, data.data() + 84 + 50ull * i + 12, 36);
        verts.insert(verts.end(), v, v + 9);
        for (uint32_t k = 0; k < 3; k++) tris.push_back(3 * i + k);
      }
      return true;
    }
  }
  size_t p = 0;
  while (p < data.size() && std::isspace((unsigned char)data[p])) p++;
  if (data.compare(p, 5, "solid") != 0) return false;
  const char* s = data.c_str();
  while ((s = std::strstr(s, "vertex")) != nullptr) {
    float x, y, z;
    if (std::sscanf(s + 6, "%f %f %f", &x, &y, &z) != 3) return false;
    verts.push_back(x); verts.push_back(y); verts.push_back(z);
    s += 6;
  }
  if ((verts.size() / 3) % 3) return false;
  for (uint32_t i = 0; i < verts.size() / 3; i++) tris.push_back(i);
  return true;
}


// ---------------------------------------------------------------------------------------------
// Collada (.dae) and Wavefront OBJ.  What of Assimp's behaviour (src/renderable.cpp:306-322 reads the file with
// PreTransformVertices | SortByPType | GenNormals | Triangulate | GenUVCoords | FlipUVs; :352-415 take positions and three
// indices per face) is restated here, and that none of it can be pinned against an Assimp in this image, is written
// down in realtime_urdf_filter_amd/meshes.py -- the Python twin of this code; both are tested against each other.
// ---------------------------------------------------------------------------------------------
struct MeshOptions {
  bool up_axis_to_y = true;     // Assimp's Collada importer rotates Z_UP / X_UP files to Y_UP (root node), PreTransformVertices bakes it in
  bool apply_unit = false;      // Assimp >= 4.1 scales the root node by <unit meter>; 3.x (Ubuntu 14.04, README.md:29) does not
};

namespace detail {
struct Mat4f {
  float m[4][4];
  static Mat4f identity() { Mat4f r{}; for (int i = 0; i < 4; i++) r.m[i][i] = 1.f; return r; }
  Mat4f operator*(const Mat4f& b) const
  {
    Mat4f r{};
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        float acc = 0.f;
        for (int k = 0; k < 4; k++) acc += m[i][k] * b.m[k][j];
        r.m[i][j] = acc;
      }
    return r;
  }
};
inline std::vector<double> parse_numbers(const std::string& s)
{
  std::vector<double> v;
  const char* p = s.c_str();
  for (;;) {
    char* e = nullptr;
    const double d = std::strtod(p, &e);
    if (e == p) break;
    v.push_back(d);
    p = e;
  }
  return v;
}
inline std::string strip_hash(const std::string& s) { return !s.empty() && s[0] == '#' ? s.substr(1) : s; }
inline void fan(const std::vector<long>& poly, std::vector<long>& corners)
{
  for (size_t k = 1; k + 1 < poly.size(); k++) { corners.push_back(poly[0]); corners.push_back(poly[k]); corners.push_back(poly[k + 1]); }
}
inline Mat4f collada_node_matrix(const XmlNode& node)
{
  Mat4f m = Mat4f::identity();
  for (const auto& e : node.children) {
    const std::vector<double> v = (e.tag == "matrix" || e.tag == "translate" || e.tag == "scale" || e.tag == "rotate") ? parse_numbers(e.text) : std::vector<double>();
    Mat4f k = Mat4f::identity();
    if (e.tag == "matrix" && v.size() == 16) {
      for (int i = 0; i < 16; i++) k.m[i / 4][i % 4] = (float)v[i];
    } else if (e.tag == "translate" && v.size() >= 3) {
      for (int i = 0; i < 3; i++) k.m[i][3] = (float)v[i];
    } else if (e.tag == "scale" && v.size() >= 3) {
      for (int i = 0; i < 3; i++) k.m[i][i] = (float)v[i];
    } else if (e.tag == "rotate" && v.size() >= 4) {
      const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      if (n <= 0) continue;
      const double x = v[0] / n, y = v[1] / n, z = v[2] / n, a = v[3] * M_PI / 180.0;
      const double c = std::cos(a), s = std::sin(a), o = 1.0 - std::cos(a);
      const double r[3][3] = {{c + x * x * o, x * y * o - z * s, x * z * o + y * s},
                              {y * x * o + z * s, c + y * y * o, y * z * o - x * s},
                              {z * x * o - y * s, z * y * o + x * s, c + z * z * o}};
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) k.m[i][j] = (float)r[i][j];
    } else {
      continue;
    }
    m = m * k;
  }
  return m;
}
struct ColladaGeometry { bool ok = false; std::vector<float> pts; std::vector<long> corners; };
inline ColladaGeometry collada_read_geometry(const XmlNode& geom)
{
  ColladaGeometry g;
  const XmlNode* mesh = geom.child("mesh");
  if (!mesh) return g;
  std::map<std::string, std::pair<std::vector<float>, int>> sources;        // id -> (values, stride)
  for (const XmlNode* src : mesh->all("source")) {
    const XmlNode* fa = src->child("float_array");
    if (!fa) continue;
    const std::vector<double> vals = parse_numbers(fa->text);
    const XmlNode* acc = src->path({"technique_common", "accessor"});
    const long stride = acc ? std::atol(acc->get("stride", "3").c_str()) : 3;
    const long offset = acc ? std::atol(acc->get("offset", "0").c_str()) : 0;
    long count = ((long)vals.size() - offset) / std::max(stride, 1L);
    if (acc && !acc->get("count").empty()) count = std::atol(acc->get("count").c_str());
    if (stride < 1 || offset < 0 || count < 0 || offset + count * stride > (long)vals.size()) throw std::runtime_error("Collada: accessor outside its float_array");
    std::vector<float> a((size_t)(count * stride));
    for (long i = 0; i < count * stride; i++) a[(size_t)i] = (float)vals[(size_t)(offset + i)];
    sources[src->get("id")] = {std::move(a), (int)stride};
  }
  const XmlNode* vtx = mesh->child("vertices");
  if (!vtx) return g;
  const std::string vertices_id = vtx->get("id");
  const std::pair<std::vector<float>, int>* pos = nullptr;
  for (const XmlNode* inp : vtx->all("input"))
    if (inp->get("semantic") == "POSITION") {
      auto it = sources.find(strip_hash(inp->get("source")));
      if (it != sources.end()) pos = &it->second;
    }
  if (!pos) return g;
  const int ps = pos->second;
  const size_t npts = pos->first.size() / (size_t)ps;
  g.pts.assign(npts * 3, 0.f);
  for (size_t i = 0; i < npts; i++)
    for (int k = 0; k < std::min(ps, 3); k++) g.pts[3 * i + k] = pos->first[i * ps + k];
  for (const auto& prim : mesh->children) {
    const std::string& t = prim.tag;
    if (t != "triangles" && t != "polylist" && t != "polygons" && t != "trifans" && t != "tristrips") continue;
    long stride = 1, voff = -1;
    for (const XmlNode* inp : prim.all("input")) {
      const long off = std::atol(inp->get("offset", "0").c_str());
      stride = std::max(stride, off + 1);
      if (inp->get("semantic") == "VERTEX" && strip_hash(inp->get("source")) == vertices_id) voff = off;
    }
    if (voff < 0) continue;
    auto vertex_indices = [&](const std::string& text) {
      const std::vector<double> p = parse_numbers(text);
      std::vector<long> out;
      for (size_t i = (size_t)voff; i < p.size(); i += (size_t)stride) out.push_back((long)p[i]);
      return out;
    };
    if (t == "triangles") {
      const XmlNode* p = prim.child("p");
      std::vector<long> idx = vertex_indices(p ? p->text : "");
      idx.resize(idx.size() / 3 * 3);
      g.corners.insert(g.corners.end(), idx.begin(), idx.end());
    } else if (t == "polylist") {
      const XmlNode *p = prim.child("p"), *vc = prim.child("vcount");
      const std::vector<long> idx = vertex_indices(p ? p->text : "");
      size_t at = 0;
      for (double nd : parse_numbers(vc ? vc->text : "")) {
        const size_t n = (size_t)nd;
        if (n >= 3 && at + n <= idx.size()) fan(std::vector<long>(idx.begin() + at, idx.begin() + at + n), g.corners);
        at += n;
      }
    } else if (t == "polygons" || t == "trifans") {
      for (const XmlNode* p : prim.all("p")) fan(vertex_indices(p->text), g.corners);
    } else {
      for (const XmlNode* p : prim.all("p")) {
        const std::vector<long> idx = vertex_indices(p->text);
        for (size_t k = 0; k + 2 < idx.size(); k++) {
          if (k % 2 == 0) { g.corners.push_back(idx[k]); g.corners.push_back(idx[k + 1]); }
          else { g.corners.push_back(idx[k + 1]); g.corners.push_back(idx[k]); }
          g.corners.push_back(idx[k + 2]);
        }
      }
    }
  }
  for (long c : g.corners) if (c < 0 || (size_t)c >= npts) throw std::runtime_error("Collada: vertex index out of range in geometry " + geom.get("id"));
  g.ok = true;
  return g;
}
inline void strip_namespace_prefixes(XmlNode& n)          // <dae:mesh> -> <mesh> (only here: URDFs carry prefixed extension tags that must stay apart)
{
  const size_t colon = n.tag.find(':');
  if (colon != std::string::npos) n.tag = n.tag.substr(colon + 1);
  for (auto& c : n.children) strip_namespace_prefixes(c);
}
inline void collect_library_nodes(const XmlNode& n, std::map<std::string, const XmlNode*>& out)
{
  for (const auto& c : n.children) {
    if (c.tag == "node") { if (!c.get("id").empty()) out[c.get("id")] = &c; }
    collect_library_nodes(c, out);
  }
}
}  // namespace detail

inline bool load_collada(const std::string& data, std::vector<float>& verts, std::vector<uint32_t>& tris, const MeshOptions& opt = MeshOptions())
{
  using namespace detail;
  verts.clear();
  tris.clear();
  XmlNode root;
  try {
    root = XmlParser(data).parse_document();
    strip_namespace_prefixes(root);
    if (root.tag != "COLLADA") return false;
    std::string up = "Y_UP";
    double meter = 1.0;
    if (const XmlNode* asset = root.child("asset")) {
      if (const XmlNode* u = asset->child("up_axis")) { up.clear(); for (char ch : u->text) if (!std::isspace((unsigned char)ch)) up.push_back((char)std::toupper((unsigned char)ch)); }
      if (const XmlNode* u = asset->child("unit")) meter = std::atof(u->get("meter", "1").c_str());
    }
    std::map<std::string, const XmlNode*> geoms, lib_nodes, scenes;
    const XmlNode* first_scene = nullptr;
    for (const XmlNode* lib : root.all("library_geometries")) for (const XmlNode* g : lib->all("geometry")) geoms[g->get("id")] = g;
    for (const XmlNode* lib : root.all("library_nodes")) collect_library_nodes(*lib, lib_nodes);
    for (const XmlNode* lib : root.all("library_visual_scenes"))
      for (const XmlNode* sc : lib->all("visual_scene")) { scenes[sc->get("id")] = sc; if (!first_scene) first_scene = sc; }
    const XmlNode* scene = nullptr;
    if (const XmlNode* inst = root.path({"scene", "instance_visual_scene"})) {
      auto it = scenes.find(strip_hash(inst->get("url")));
      if (it != scenes.end()) scene = it->second;
    }
    if (!scene) scene = first_scene;
    Mat4f rootm = Mat4f::identity();
    if (opt.up_axis_to_y && up == "Z_UP") { Mat4f k{}; k.m[0][0] = 1; k.m[1][2] = 1; k.m[2][1] = -1; k.m[3][3] = 1; rootm = rootm * k; }
    else if (opt.up_axis_to_y && up == "X_UP") { Mat4f k{}; k.m[0][1] = -1; k.m[1][0] = 1; k.m[2][2] = 1; k.m[3][3] = 1; rootm = rootm * k; }
    if (opt.apply_unit) { Mat4f k = Mat4f::identity(); for (int i = 0; i < 3; i++) k.m[i][i] = (float)meter; rootm = rootm * k; }
    std::map<std::string, ColladaGeometry> cache;
    auto emit = [&](const std::string& gid, const Mat4f& m) {
      auto it = cache.find(gid);
      if (it == cache.end()) {
        auto g = geoms.find(gid);
        it = cache.emplace(gid, g == geoms.end() ? ColladaGeometry() : collada_read_geometry(*g->second)).first;
      }
      const ColladaGeometry& g = it->second;
      if (!g.ok) return;
      for (long c : g.corners) {
        const float x = g.pts[3 * c], y = g.pts[3 * c + 1], z = g.pts[3 * c + 2];
        for (int r = 0; r < 3; r++) verts.push_back(m.m[r][0] * x + m.m[r][1] * y + m.m[r][2] * z + m.m[r][3]);
      }
    };
    std::function<void(const XmlNode&, const Mat4f&, int)> walk = [&](const XmlNode& node, const Mat4f& parent, int depth) {
      if (depth > 64) throw std::runtime_error("Collada: node hierarchy too deep");
      const Mat4f m = parent * collada_node_matrix(node);
      for (const auto& e : node.children) {
        if (e.tag == "instance_geometry") emit(strip_hash(e.get("url")), m);
        else if (e.tag == "instance_node") { auto it = lib_nodes.find(strip_hash(e.get("url"))); if (it != lib_nodes.end()) walk(*it->second, m, depth + 1); }
        else if (e.tag == "node") walk(e, m, depth + 1);
      }
    };
    if (scene) { for (const XmlNode* n : scene->all("node")) walk(*n, rootm, 0); }
    else { for (const auto& g : geoms) emit(g.first, rootm); }
  } catch (const std::exception&) {
    verts.clear();
    return false;
  }
  if (verts.empty()) return false;
  for (uint32_t i = 0; i < verts.size() / 3; i++) tris.push_back(i);
  return true;
}

inline bool load_obj(const std::string& data, std::vector<float>& verts, std::vector<uint32_t>& tris)
{
  verts.clear();
  tris.clear();
  std::vector<float> pts;
  std::string text = data;
  for (size_t p = 0; (p = text.find("\\\n", p)) != std::string::npos;) text.replace(p, 2, " ");
  size_t at = 0;
  while (at < text.size()) {
    size_t e = text.find('\n', at);
    if (e == std::string::npos) e = text.size();
    std::string line = text.substr(at, e - at);
    at = e + 1;
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line.resize(hash);
    std::vector<std::string> tok;
    for (size_t p = 0; p < line.size();) {
      while (p < line.size() && std::isspace((unsigned char)line[p])) p++;
      size_t q = p;
      while (q < line.size() && !std::isspace((unsigned char)line[q])) q++;
      if (q > p) tok.push_back(line.substr(p, q - p));
      p = q;
    }
    if (tok.empty()) continue;
    if (tok[0] == "v" && tok.size() >= 4) {
      for (int k = 1; k <= 3; k++) pts.push_back((float)std::atof(tok[k].c_str()));
    } else if (tok[0] == "f" && tok.size() >= 4) {
      std::vector<long> poly, corners;
      const long n = (long)(pts.size() / 3);
      for (size_t k = 1; k < tok.size(); k++) {
        long i = std::atol(tok[k].c_str());            // stops at the first '/'
        i = i > 0 ? i - 1 : n + i;
        if (i < 0 || i >= n) { verts.clear(); return false; }
        poly.push_back(i);
      }
      detail::fan(poly, corners);
      for (long c : corners) for (int k = 0; k < 3; k++) verts.push_back(pts[3 * c + k]);
    }
  }
  if (pts.empty()) return false;
  for (uint32_t i = 0; i < verts.size() / 3; i++) tris.push_back(i);
  return true;
}

// By extension of `name` (file name or URI), else by content.
inline bool load_mesh(const std::string& name, const std::string& data, std::vector<float>& verts, std::vector<uint32_t>& tris,
                      const MeshOptions& opt = MeshOptions())
{
  std::string ext;
  const size_t dot = name.rfind('.');
  if (dot != std::string::npos) for (size_t i = dot + 1; i < name.size(); i++) ext.push_back((char)std::tolower((unsigned char)name[i]));
  if (ext == "dae") return load_collada(data, verts, tris, opt);
  if (ext == "obj") return load_obj(data, verts, tris);
  if (ext == "stl" || ext == "stlb" || ext == "stla") return load_stl(data, verts, tris);
  size_t p = 0;
  while (p < data.size() && std::isspace((unsigned char)data[p])) p++;
  const std::string head = data.substr(p, 512);
  if (head.compare(0, 5, "<?xml") == 0 || head.find("<COLLADA") != std::string::npos) return load_collada(data, verts, tris, opt);
  if (load_stl(data, verts, tris)) return true;
  return load_obj(data, verts, tris);
}

// resource_retriever's job for the façade when the host supplies no MeshResolver: package://<pkg>/<path> against the
// roots in ROS_PACKAGE_PATH (a root either is the package directory or contains it), file://<path>, plain paths.
inline bool default_mesh_resolver(const std::string& uri, std::string& data, void* /*user*/)
{
  auto slurp = [&](const std::string& path) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    data.clear();
    char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, n);
    std::fclose(f);
    return true;
  };
  if (uri.compare(0, 10, "package://") == 0) {
    const size_t slash = uri.find('/', 10);
    if (slash == std::string::npos) return false;
    const std::string pkg = uri.substr(10, slash - 10), rel = uri.substr(slash + 1);
    const char* env = std::getenv("ROS_PACKAGE_PATH");
    std::string roots = env ? env : "";
    for (size_t at = 0; at <= roots.size();) {
      size_t e = roots.find(':', at);
      if (e == std::string::npos) e = roots.size();
      std::string root = roots.substr(at, e - at);
      at = e + 1;
      if (root.empty()) continue;
      while (root.size() > 1 && root.back() == '/') root.pop_back();
      if (slurp(root + "/" + pkg + "/" + rel)) return true;
      const size_t base = root.rfind('/');
      if ((base == std::string::npos ? root : root.substr(base + 1)) == pkg && slurp(root + "/" + rel)) return true;
    }
    return false;
  }
  if (uri.compare(0, 7, "file://") == 0) return slurp(uri.substr(7));
  return slurp(uri);
}

}  // namespace rtuf_host
