// scene.cpp — dependency-free re-implementation of the reference loaders and of
// the small host-side methods of Camera / Triangle / Material.
// Format semantics follow scene.cpp:3-213, camera.cpp:3-28, triangle.cpp:3-29,
// material.cpp:3-11 of the reference.
#include "scene.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace trt {

// ---------------------------------------------------------------- Camera ----
void Camera::setCamera()
{
    // camera.cpp:3-17: the half-height is evaluated in double, the viewport
    // extents are narrowed to float before they scale the basis vectors.
    const double theta = fovy * 0.01745329251994329576923690768489;
    const double h = std::tan(theta / 2);
    const float viewport_height = (float)(2.0 * h);
    const float viewport_width = (float)(aspect_ratio * viewport_height);

    const vec3 w = normalize(eye - lookat);
    const vec3 u = normalize(cross(up, w));
    const vec3 v = cross(w, u);

    horizontal = viewport_width * u;
    vertical = viewport_height * v;
    lower_left_corner = eye - horizontal / 2.0f - vertical / 2.0f - w;
}

Ray Camera::getRay(float s, float t) const
{
    // camera.cpp:19-28
    vec3 d = lower_left_corner + s * horizontal + t * vertical - eye;
    return Ray(eye, normalize(d));
}

void Camera::Print() const
{
    std::printf("Camera:\nfovy: %f eye: (%f, %f, %f) lookat: (%f, %f, %f) up: (%f, %f, %f)\n", fovy,
                eye.x, eye.y, eye.z, lookat.x, lookat.y, lookat.z, up.x, up.y, up.z);
}

// -------------------------------------------------------------- Triangle ----
double Triangle::calAera() const
{
    // triangle.cpp:3-10: law of cosines on float edge lengths, in double.
    const double a = length(v[1] - v[0]), b = length(v[2] - v[0]), c = length(v[2] - v[1]);
    const double cos_c = (a * a + b * b - c * c) / (2 * a * b);
    const double sin_c = std::sqrt(1 - cos_c * cos_c);
    return a * b * sin_c / 2;
}

vec3 Triangle::findBaryCor(vec3 p) const
{
    const double e1[3] = {(double)v[1].x - v[0].x, (double)v[1].y - v[0].y, (double)v[1].z - v[0].z};
    const double e2[3] = {(double)v[2].x - v[0].x, (double)v[2].y - v[0].y, (double)v[2].z - v[0].z};
    const double q[3] = {(double)p.x - v[0].x, (double)p.y - v[0].y, (double)p.z - v[0].z};
    auto dt = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    const double d00 = dt(e1, e1), d01 = dt(e1, e2), d11 = dt(e2, e2), d20 = dt(q, e1), d21 = dt(q, e2);
    const double den = d00 * d11 - d01 * d01;
    const double b1 = (d11 * d20 - d01 * d21) / den;
    const double b2 = (d00 * d21 - d01 * d20) / den;
    return vec3((float)(1.0 - b1 - b2), (float)b1, (float)b2);
}

// -------------------------------------------------------------- Material ----
static bool readPPM(const std::string& path, std::vector<uint8_t>& rgb, int& w, int& h)
{
    FILE* fp = std::fopen(path.c_str(), "rb");
    if (!fp) return false;
    auto token = [&](std::string& out) {
        out.clear();
        int c = std::fgetc(fp);
        for (;;) {
            while (c == ' ' || c == '\n' || c == '\r' || c == '\t') c = std::fgetc(fp);
            if (c == '#') { while (c != '\n' && c != EOF) c = std::fgetc(fp); continue; }
            break;
        }
        while (c != EOF && c != ' ' && c != '\n' && c != '\r' && c != '\t') { out.push_back((char)c); c = std::fgetc(fp); }
        return !out.empty();
    };
    std::string magic, sw, sh, smax;
    bool ok = token(magic) && magic == "P6" && token(sw) && token(sh) && token(smax);
    if (ok) {
        w = std::atoi(sw.c_str());
        h = std::atoi(sh.c_str());
        ok = w > 0 && h > 0 && std::atoi(smax.c_str()) == 255;
    }
    if (ok) {
        rgb.resize((size_t)w * h * 3);
        ok = std::fread(rgb.data(), 1, rgb.size(), fp) == rgb.size();
    }
    std::fclose(fp);
    if (!ok) rgb.clear();
    return ok;
}

bool decodeJPEG(const std::string& path, std::vector<uint8_t>& rgb, int& width, int& height);  // jpeg.cpp
bool decodePNG(const std::string& path, std::vector<uint8_t>& rgb, int& width, int& height);   // png.cpp

bool Material::readinMap()
{
    // material.cpp:3-11 decodes with cv::imread.  OpenCV/libjpeg/libpng are not available: binary PPM is read
    // directly, baseline and progressive JPEG through host/jpeg.cpp (libjpeg's integer IDCT / fancy upsampling / colour
    // tables restated, so the texels are the ones cv::imread yields), PNG through host/png.cpp (every colour type
    // and depth, interlaced or not, reduced to 8-bit RGB the way imread's default flag does), and as a last resort
    // the pre-decoded sidecar "<map_Kd>.ppm" written by tools/decode_textures.py (CMYK or arithmetic-coded JPEGs, ...).
    img.clear();
    map_width = map_height = 0;
    if (readPPM(map_Kd, img, map_width, map_height)) return true;
    if (decodeJPEG(map_Kd, img, map_width, map_height)) return true;
    img.clear();
    if (decodePNG(map_Kd, img, map_width, map_height)) return true;
    img.clear();
    if (readPPM(map_Kd + ".ppm", img, map_width, map_height)) return true;
    std::printf("Cannot read file: %s\n", map_Kd.c_str());
    return false;
}

// ------------------------------------------------------------------ Scene ----
Material& Scene::material(const std::string& name) { return materials[(size_t)materialId(name)]; }

int Scene::materialId(const std::string& name)
{
    auto it = material_ids.find(name);
    if (it != material_ids.end()) return it->second;
    const int id = (int)materials.size();
    materials.emplace_back();
    materials.back().name = name;
    material_ids.emplace(name, id);
    return id;
}

void Scene::setResolution(int width, int height)
{
    if (width <= 0 || height <= 0) throw std::runtime_error("setResolution: non-positive size");
    img_width = width;
    img_height = height;
    camera.aspect_ratio = (double)img_width / (double)img_height;
    camera.setCamera();
}

namespace {

// One element of the scene description: name + attributes.  The format is a
// sequence of top-level elements (camera with three children, then lights), so
// a flat tag scan in document order is all the structure that is needed.
struct XmlTag {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attrs;
    const std::string* get(const char* key) const
    {
        for (auto& kv : attrs)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

std::vector<XmlTag> scanXml(const std::string& text)
{
    std::vector<XmlTag> tags;
    size_t i = 0;
    const size_t n = text.size();
    auto isname = [](char c) { return std::isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.'; };
    while (i < n) {
        if (text[i] != '<') { ++i; continue; }
        if (text.compare(i, 4, "<!--") == 0) {
            size_t e = text.find("-->", i + 4);
            i = (e == std::string::npos) ? n : e + 3;
            continue;
        }
        if (i + 1 < n && (text[i + 1] == '?' || text[i + 1] == '!' || text[i + 1] == '/')) {
            size_t e = text.find('>', i);
            i = (e == std::string::npos) ? n : e + 1;
            continue;
        }
        ++i;
        XmlTag tag;
        while (i < n && isname(text[i])) tag.name.push_back(text[i++]);
        for (;;) {
            while (i < n && std::isspace((unsigned char)text[i])) ++i;
            if (i >= n) throw std::runtime_error("unterminated element <" + tag.name);
            if (text[i] == '>') { ++i; break; }
            if (text[i] == '/') { ++i; continue; }
            std::string key;
            while (i < n && isname(text[i])) key.push_back(text[i++]);
            if (key.empty()) throw std::runtime_error("malformed attribute in <" + tag.name + ">");
            while (i < n && std::isspace((unsigned char)text[i])) ++i;
            if (i >= n || text[i] != '=') throw std::runtime_error("attribute without value in <" + tag.name + ">");
            ++i;
            while (i < n && std::isspace((unsigned char)text[i])) ++i;
            if (i >= n || (text[i] != '"' && text[i] != '\'')) throw std::runtime_error("unquoted attribute in <" + tag.name + ">");
            const char q = text[i++];
            std::string val;
            while (i < n && text[i] != q) val.push_back(text[i++]);
            if (i >= n) throw std::runtime_error("unterminated attribute in <" + tag.name + ">");
            ++i;
            tag.attrs.emplace_back(std::move(key), std::move(val));
        }
        tags.push_back(std::move(tag));
    }
    return tags;
}

float attrFloat(const XmlTag& t, const char* key)
{
    const std::string* s = t.get(key);
    if (!s) throw std::runtime_error("<" + t.name + "> lacks attribute " + key);
    char* end = nullptr;
    const float v = std::strtof(s->c_str(), &end);  // like stof: leading blanks skipped, prefix parsed
    if (end == s->c_str()) throw std::runtime_error("<" + t.name + " " + key + "> is not a number");
    return v;
}

vec3 attrXYZ(const XmlTag& t) { return vec3(attrFloat(t, "x"), attrFloat(t, "y"), attrFloat(t, "z")); }

std::string slurp(const std::string& path, const char* what)
{
    std::ifstream fin(path, std::ios::binary);
    if (!fin.is_open()) throw std::runtime_error(std::string("Read ") + what + " failed: " + path);
    std::ostringstream ss;
    ss << fin.rdbuf();
    return ss.str();
}

}  // namespace

void Scene::readxml(const std::string& xml_path)
{
    const std::vector<XmlTag> tags = scanXml(slurp(xml_path, "xml"));
    size_t i = 0;
    while (i < tags.size() && tags[i].name != "camera") ++i;
    if (i == tags.size()) throw std::runtime_error("Read xml failed: no <camera> in " + xml_path);
    const XmlTag& cam = tags[i];
    img_width = (int)attrFloat(cam, "width");
    img_height = (int)attrFloat(cam, "height");
    camera.aspect_ratio = (double)img_width / (double)img_height;
    camera.fovy = attrFloat(cam, "fovy");  // stof -> float -> double, scene.cpp:16
    bool have_eye = false, have_lookat = false, have_up = false;
    for (const XmlTag& t : tags) {
        if (t.name == "eye" && !have_eye) { camera.eye = attrXYZ(t); have_eye = true; }
        else if (t.name == "lookat" && !have_lookat) { camera.lookat = attrXYZ(t); have_lookat = true; }
        else if (t.name == "up" && !have_up) { camera.up = attrXYZ(t); have_up = true; }
    }
    if (!have_eye || !have_lookat || !have_up) throw std::runtime_error("Read xml failed: camera lacks eye/lookat/up");
    camera.setCamera();

    for (size_t k = i + 1; k < tags.size(); ++k) {
        const XmlTag& t = tags[k];
        if (t.name != "light") continue;
        const std::string* name = t.get("mtlname");
        const std::string* rad = t.get("radiance");
        if (!name || !rad) throw std::runtime_error("Read xml failed: <light> lacks mtlname/radiance");
        // "r,g,b": split on the first two commas, each part parsed like stof
        // (blanks and newlines after a comma are legal, staircase.xml:10-12).
        vec3 radiance;
        const size_t c1 = rad->find(',');
        const size_t c2 = (c1 == std::string::npos) ? std::string::npos : rad->find(',', c1 + 1);
        if (c2 == std::string::npos) throw std::runtime_error("Read xml failed: radiance needs three components");
        radiance.x = std::strtof(rad->substr(0, c1).c_str(), nullptr);
        radiance.y = std::strtof(rad->substr(c1 + 1, c2 - c1 - 1).c_str(), nullptr);
        radiance.z = std::strtof(rad->substr(c2 + 1).c_str(), nullptr);
        lights.push_back(Light(*name, radiance));
        Material& m = material(*name);
        m.is_emissive = true;
        m.radiance = radiance;
    }
}

void Scene::readmtl(const std::string& mtl_path, const std::string& basedir)
{
    std::ifstream fin(mtl_path);
    if (!fin.is_open()) throw std::runtime_error("Read " + mtl_path + " failed.");
    std::string line, current;
    while (std::getline(fin, line)) {
        std::istringstream sin(line);
        std::string key;
        sin >> key;
        if (key == "newmtl") {
            sin >> current;
            materialId(current);
        } else if (key == "Kd" || key == "Ks" || key == "Tr") {
            float x = 0, y = 0, z = 0;
            sin >> x >> y >> z;
            Material& m = material(current);
            (key == "Kd" ? m.Kd : key == "Ks" ? m.Ks : m.Tr) = vec3(x, y, z);
        } else if (key == "Ns" || key == "Ni") {
            float v = 0;
            sin >> v;
            Material& m = material(current);
            (key == "Ns" ? m.Ns : m.Ni) = v;
        } else if (key == "map_Kd") {
            std::string rel;
            sin >> rel;
            Material& m = material(current);
            m.map_Kd = basedir + "/" + rel;
            m.readinMap();
        }
        // anything else (Kt, illum, comments...) is ignored, as in the reference
    }
}

namespace {

// Parses one face-vertex token.  The reference accepts only "a/b/c" with
// positive indices (scene.cpp:165-195); this also takes "a//c", "a/b", "a" and
// negative (relative) indices.  Missing slots come back as 0.
void parseFaceToken(const std::string& tok, long idx[3])
{
    idx[0] = idx[1] = idx[2] = 0;
    int slot = 0;
    size_t start = 0;
    for (size_t i = 0; i <= tok.size() && slot < 3; ++i) {
        if (i == tok.size() || tok[i] == '/') {
            if (i > start) idx[slot] = std::strtol(tok.substr(start, i - start).c_str(), nullptr, 10);
            ++slot;
            start = i + 1;
        }
    }
}

template <class T>
const T& pick(const std::vector<T>& arr, long one_based, const char* what)
{
    long i = one_based > 0 ? one_based - 1 : (long)arr.size() + one_based;
    if (one_based == 0 || i < 0 || i >= (long)arr.size()) throw std::runtime_error(std::string("obj: ") + what + " index out of range");
    return arr[(size_t)i];
}

}  // namespace

void Scene::readobj(const std::string& obj_path)
{
    std::ifstream fin(obj_path);
    if (!fin.is_open()) throw std::runtime_error("Read " + obj_path + " failed.");

    std::vector<vec3> vertices, vn;
    std::vector<vec2> vt;
    // Slot order of face tokens (scene.cpp:149-152,175-191): if a `vt` line is
    // seen before any `vn` line the tokens are v/vt/vn (the OBJ standard),
    // otherwise v/vn/vt.
    bool second_slot_is_vn = true;
    std::string mtl_name;
    int mtl_id = -1;
    std::string line;
    while (std::getline(fin, line)) {
        std::istringstream sin(line);
        std::string key;
        sin >> key;
        if (key == "v") {
            float x = 0, y = 0, z = 0;
            sin >> x >> y >> z;
            vertices.push_back(vec3(x, y, z));
        } else if (key == "vn") {
            float x = 0, y = 0, z = 0;
            sin >> x >> y >> z;
            vn.push_back(vec3(x, y, z));
        } else if (key == "vt") {
            if (vn.empty()) second_slot_is_vn = false;
            float x = 0, y = 0;
            sin >> x >> y;
            vt.push_back(vec2(x, y));
        } else if (key == "usemtl") {
            sin >> mtl_name;
            mtl_id = materialId(mtl_name);
        } else if (key == "f") {
            // Only the first three vertex tokens are used (scene.cpp:162) — unless triangulate_polygons asks for the fan.
            std::vector<std::string> toks;
            for (std::string t; sin >> t;) toks.push_back(t);
            if (toks.size() < 3) throw std::runtime_error("obj: face with fewer than three vertices");
            if (mtl_id < 0) mtl_id = materialId(mtl_name);
            const size_t n_fan = (triangulate_polygons && toks.size() > 3) ? toks.size() - 2 : 1;
            for (size_t f = 0; f < n_fan; ++f) {
            const std::string* tok3[3] = {&toks[0], &toks[f + 1], &toks[f + 2]};
            Triangle tri;
            bool have_vn = true;
            for (int k = 0; k < 3; ++k) {
                long idx[3];
                parseFaceToken(*tok3[k], idx);
                tri.v[k] = pick(vertices, idx[0], "vertex");
                // The slot-order quirk applies to the full a/b/c form the reference parses; the
                // forms it cannot parse follow the OBJ standard: a//c = v//vn, a/b = v/vt.
                const bool full = idx[1] != 0 && idx[2] != 0;
                const long i_vn = (full && second_slot_is_vn) ? idx[1] : idx[2];
                const long i_vt = (full && second_slot_is_vn) ? idx[2] : idx[1];
                if (i_vn != 0) tri.vn[k] = pick(vn, i_vn, "normal"); else have_vn = false;
                if (i_vt != 0) tri.vt[k] = pick(vt, i_vt, "texcoord");
            }
            tri.normal = normalize(cross(tri.v[1] - tri.v[0], tri.v[2] - tri.v[0]));
            if (!have_vn) tri.vn[0] = tri.vn[1] = tri.vn[2] = tri.normal;
            tri.center = (tri.v[0] + tri.v[1] + tri.v[2]) / 3.0f;
            tri.mtl_name = mtl_name;
            tri.mtl_id = mtl_id;
            Material& m = materials[(size_t)mtl_id];
            if (m.is_emissive) {
                // scene.cpp:199-205: running total area = the light's CDF
                tri.is_emissive = true;
                m.area += tri.calAera();
                tri.area = m.area;
                m.triangles.push_back(tri);
            }
            triangles.push_back(std::move(tri));
            }
        }
    }
    n_vertices = (int)vertices.size();
    n_vn = (int)vn.size();
    n_vt = (int)vt.size();
}

}  // namespace trt
