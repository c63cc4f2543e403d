"""rtbhip.xacro -- a small xacro expander (stdlib only): robot descriptions written in the ROS xacro macro language -> plain URDF text.

Why it is here: the reference reads its robot models from xacro files (`Robot.URDF_read` robot/Robot.py:218-286 hands `*.xacro` paths to its bundled
`tools/xacro`), so `Robot.URDF("…/panda_arm_hand.urdf.xacro")` is part of the data format in front of the hot path.  This module is an
independent implementation of the xacro LANGUAGE as its documentation describes it (wiki.ros.org/xacro) -- properties (lazy, typed, with blocks),
`${…}` expressions, `$(arg …)` / `$(find …)` / `$(optenv …)` substitutions, macros with default / forwarded / block parameters, `insert_block`,
`if` / `unless`, `include`, `arg` -- sized for the descriptions the reference ships (rtb-data/rtbdata/xacro/**); what it does not implement
(`xacro:element` / `xacro:attribute`, `load_yaml`, include namespaces) raises XacroError instead of guessing.  Host-side text processing: nothing
here touches the device.

    urdf_text = rtbhip.xacro.process("/path/to/robot.urdf.xacro", mappings={"arm_id": "panda"}, packages=["/path/to/the/description/folders"])
"""
import copy
import math
import os
import re
import xml.etree.ElementTree as ET


class XacroError(ValueError):
    pass


def _is_xacro_ns(uri):
    return "xacro" in uri


def _split(tag):
    """('xacro' | '' | other-namespace, local name) of an ElementTree tag"""
    if isinstance(tag, str) and tag.startswith("{"):
        uri, name = tag[1:].split("}", 1)
        return ("xacro" if _is_xacro_ns(uri) else uri), name
    return "", tag


def _literal(value):
    """A property / parameter value given as text: a quoted string, an int, a float, a boolean -- else the text itself."""
    if isinstance(value, str):
        if len(value) >= 2 and value[0] == "'" and value[-1] == "'":
            return value[1:-1]
        for conv in (int, float):
            try:
                return conv(value)
            except ValueError:
                pass
        if value.strip().lower() in ("true", "false"):
            return value.strip().lower() == "true"
    return value


def _boolean(value, what):
    if isinstance(value, bool):
        return value
    if isinstance(value, (int, float)):
        return bool(value)
    if isinstance(value, str):
        v = value.strip().lower()
        if v in ("true", "false"):
            return v == "true"
        try:
            return bool(float(v))
        except ValueError:
            pass
    raise XacroError("%s: %r is not a boolean" % (what, value))


class _Block:
    """A property / macro argument that is XML, not a value: `elements` is what an insert_block puts in its place."""

    def __init__(self, elements, evaluated):
        self.elements, self.evaluated = elements, evaluated


class _Lazy:
    def __init__(self, text):
        self.text = text


class _Scope:
    """Symbol table with a parent; property texts are evaluated on first use, in the scope that defined them."""

    def __init__(self, ctx, parent=None):
        self.ctx, self.parent, self.d, self._busy = ctx, parent, {}, set()

    def root(self):
        s = self
        while s.parent is not None:
            s = s.parent
        return s

    def define(self, name, value):
        self.d[name] = value

    def owner(self, name):
        s = self
        while s is not None:
            if name in s.d:
                return s
            s = s.parent
        return None

    def __contains__(self, name):
        return self.owner(name) is not None

    def __getitem__(self, name):           # the mapping protocol eval() needs
        s = self.owner(name)
        if s is None:
            raise KeyError(name)
        v = s.d[name]
        if isinstance(v, _Lazy):
            if name in s._busy:
                raise XacroError("property %r is defined in terms of itself" % name)
            s._busy.add(name)
            try:
                v = _literal(s.ctx.text(v.text, s))
            finally:
                s._busy.discard(name)
            s.d[name] = v
        return v


_GLOBALS = {k: getattr(math, k) for k in ("pi", "e", "sin", "cos", "tan", "asin", "acos", "atan", "atan2", "sqrt", "radians", "degrees", "floor", "ceil",
                                          "fabs", "log", "log10", "exp", "pow", "hypot", "fmod", "copysign", "isnan", "isinf")}
_GLOBALS.update({"abs": abs, "min": min, "max": max, "round": round, "int": int, "float": float, "str": str, "bool": bool, "len": len, "list": list,
                 "dict": dict, "tuple": tuple, "range": range, "sorted": sorted, "sum": sum, "True": True, "False": False, "None": None,
                 "true": True, "false": False, "__builtins__": {}})


class _Macro:
    def __init__(self, name, params, body):
        self.name, self.body = name, body
        self.params = []                      # (name, kind, has_default, default_text, forward)   kind: '' value, '*' block, '**' spread block
        for tok in re.findall(r"[^\s:=]+\s*:=\s*(?:'[^']*'|\S+)|\S+", params or ""):
            default, has = None, False
            if ":=" in tok:
                tok, default = [x.strip() for x in tok.split(":=", 1)]
                has = True
            kind = "**" if tok.startswith("**") else ("*" if tok.startswith("*") else "")
            nm = tok[len(kind):]
            forward = False
            if has and default.startswith("^"):            # a:=^  forwards the caller's `a`; a:=^|3 falls back to 3
                forward = True
                default = default[2:] if default.startswith("^|") else None
                has = default is not None
            self.params.append((nm, kind, has, default, forward))


class _Context:
    def __init__(self, mappings, packages):
        self.args = dict(mappings or {})
        self.packages = packages
        self.files = []                        # stack of files being processed (for relative includes, $(dirname), messages)
        self.macros = {}
        self.guessed = []                      # (package name, folder) pairs find() had to guess

    # ------------------------------------------------------------------ substitution
    def find(self, pkg):
        if isinstance(self.packages, dict) and pkg in self.packages:
            return os.fspath(self.packages[pkg])
        roots = [] if self.packages is None or isinstance(self.packages, dict) else [os.fspath(p) for p in self.packages]
        for r in roots:
            if os.path.isdir(os.path.join(r, pkg)):
                return os.path.join(r, pkg)
            if os.path.basename(os.path.normpath(r)) == pkg:
                return r
        d = os.path.dirname(os.path.abspath(self.files[-1])) if self.files else os.getcwd()
        while True:                            # the description folders usually sit side by side: look up the tree of the current file
            if os.path.basename(d) == pkg:
                return d
            if os.path.isdir(os.path.join(d, pkg)):
                return os.path.join(d, pkg)
            up = os.path.dirname(d)
            if up == d:
                break
            d = up
        # A name no folder carries (descriptions copied out of their ROS workspace keep the old package names: the reference's data has
        # `$(find kuka_lbr_iiwa_support)` inside kuka_description/kuka_lbr_iiwa/): the package the CURRENT file lives in -- the parent of its
        # urdf/ or robots/ folder -- is the only candidate left; the include that follows fails loudly if the guess is wrong.
        if self.files:
            here = os.path.dirname(os.path.abspath(self.files[-1]))
            guess = os.path.dirname(here) if os.path.basename(here) in ("urdf", "robots", "xacro", "launch") else here
            self.guessed.append((pkg, guess))
            return guess
        raise XacroError("$(find %s): no such package folder (pass packages=[...] or {name: folder})" % pkg)

    def command(self, body, scope):
        parts = body.split()
        if not parts:
            raise XacroError("empty $() substitution")
        cmd, rest = parts[0], parts[1:]
        if cmd == "arg":
            if len(rest) != 1:
                raise XacroError("$(arg) takes one name")
            if rest[0] not in self.args:
                raise XacroError("undefined substitution argument %r" % rest[0])
            return self.args[rest[0]]
        if cmd == "find":
            return self.find(rest[0])
        if cmd == "env":
            if rest[0] not in os.environ:
                raise XacroError("environment variable %r is not set" % rest[0])
            return os.environ[rest[0]]
        if cmd == "optenv":
            return os.environ.get(rest[0], " ".join(rest[1:]))
        if cmd == "dirname":
            return os.path.dirname(os.path.abspath(self.files[-1]))
        if cmd == "eval":
            return self.expr(body[len("eval"):].strip(), scope)
        raise XacroError("unsupported substitution $(%s ...)" % cmd)

    def expr(self, code, scope):
        if "load_yaml" in code:
            raise XacroError("load_yaml is not supported by rtbhip.xacro")
        # expressions are evaluated with the builtins withheld; dunder attributes (`().__class__.__base__ ...`) are the way back to them.  Python
        # NFKC-normalises identifiers (fullwidth U+FF3F underscores resolve to `_`), so the check runs on the normalised text AND on the parsed
        # tree: no attribute or name may start with an underscore.  (A .xacro file is still a program: its expressions are executed.)
        import ast
        import unicodedata
        if "__" in unicodedata.normalize("NFKC", code):
            raise XacroError("double underscores are not allowed in ${%s}" % code)
        try:
            tree = ast.parse(code.strip(), mode="eval")
        except SyntaxError as e:
            raise XacroError("cannot evaluate ${%s}: SyntaxError: %s" % (code, e))
        for node in ast.walk(tree):
            ident = node.attr if isinstance(node, ast.Attribute) else node.id if isinstance(node, ast.Name) else None
            if ident is not None and unicodedata.normalize("NFKC", ident).startswith("_"):
                raise XacroError("names starting with an underscore are not allowed in ${%s}" % code)
        args = self.args

        class _Args(dict):                     # `arg('name')` inside expressions
            def __call__(self, name):
                return _literal(args[name])
        g = dict(_GLOBALS)
        g["arg"] = _Args()
        try:
            return eval(code, g, scope)        # noqa: S307 -- the language IS Python expressions; builtins are withheld
        except XacroError:
            raise
        except NameError as e:
            raise XacroError("%s in ${%s}" % (e, code))
        except Exception as e:
            raise XacroError("cannot evaluate ${%s}: %s: %s" % (code, type(e).__name__, e))

    def text(self, s, scope):
        """Text with `$$`, `${expr}`, `$(cmd)` resolved.  Text that is exactly one expression keeps the expression's type."""
        if s is None or "$" not in s:
            return s
        parts, i, n = [], 0, len(s)
        while i < n:
            j = s.find("$", i)
            if j < 0 or j == n - 1:
                parts.append(s[i:])
                break
            parts.append(s[i:j])
            c = s[j + 1]
            if c == "$" and j + 2 < n and s[j + 2] in "{(":          # $${ and $$( are escapes
                parts.append("$" + s[j + 2])
                i = j + 3
            elif c in "{(":
                close = "}" if c == "{" else ")"
                depth, k = 1, j + 2
                while k < n and depth:
                    if s[k] == c:
                        depth += 1
                    elif s[k] == close:
                        depth -= 1
                    k += 1
                if depth:
                    raise XacroError("unterminated $%s in %r" % (c, s))
                body = s[j + 2:k - 1]
                if c == "{":
                    parts.append(self.expr(self.text(body, scope) if "$(" in body else body, scope))
                else:
                    parts.append(self.command(self.text(body, scope), scope))
                i = k
            else:
                parts.append("$")
                i = j + 1
        parts = [p for p in parts if not (isinstance(p, str) and p == "")]
        if len(parts) == 1:
            return parts[0]
        return "".join(p if isinstance(p, str) else _fmt(p) for p in parts)

    # ------------------------------------------------------------------ tree walk
    def load(self, path):
        try:
            return ET.parse(path).getroot()
        except ET.ParseError as e:
            raise XacroError("%s: %s" % (path, e))

    def children(self, parent, scope):
        """The processed replacement of parent's children: [(element, )...] with texts handled by the caller."""
        out = []
        for child in list(parent):
            out.extend(self.node(child, scope))
        return out

    def node(self, el, scope):
        ns, name = _split(el.tag)
        if not isinstance(el.tag, str):                        # comments / processing instructions (only present with a custom parser)
            return [el]
        if ns != "xacro":
            new = ET.Element(el.tag)
            for k, v in el.attrib.items():
                kns, kname = _split(k)
                if kns == "xacro":
                    raise XacroError("unsupported attribute xacro:%s" % kname)
                new.set(k, _fmt(self.text(v, scope)))
            new.text = _fmt_text(self.text(el.text, scope))
            new.tail = el.tail
            for c in self.children(el, scope):
                new.append(c)
            return [new]
        attr = el.attrib
        if name == "property":
            pname = attr.get("name")
            if not pname:
                raise XacroError("xacro:property needs a name")
            target = scope
            sc = attr.get("scope")
            if sc == "parent":
                target = scope.parent or scope
            elif sc == "global":
                target = scope.root()
            if "default" in attr and "value" not in attr:
                if pname in scope:
                    return []
                target.define(pname, _Lazy(attr["default"]))
            elif "value" in attr:
                target.define(pname, _Lazy(attr["value"]))
                if sc in ("parent", "global"):
                    target.d[pname] = _literal(self.text(attr["value"], scope))        # evaluated where it was written
            else:
                target.define(pname, _Block([copy.deepcopy(c) for c in el], evaluated=False))      # a property block: its children
            return []
        if name == "arg":
            an = attr.get("name")
            if an not in self.args:
                if "default" not in attr:
                    raise XacroError("substitution argument %r has no value" % an)
                self.args[an] = _fmt(self.text(attr["default"], scope))
            return []
        if name == "macro":
            mname = attr.get("name", "")
            if mname.startswith("xacro:"):
                mname = mname[6:]
            if not mname:
                raise XacroError("xacro:macro needs a name")
            self.macros[mname] = _Macro(mname, attr.get("params", ""), copy.deepcopy(el))
            return []
        if name in ("if", "unless"):
            if "value" not in attr:
                raise XacroError("xacro:%s needs a value" % name)
            cond = _boolean(self.text(attr["value"], scope), "xacro:%s value=%r" % (name, attr["value"]))
            return self.children(el, scope) if cond == (name == "if") else []
        if name == "include":
            fn = _fmt(self.text(attr.get("filename", ""), scope))
            if attr.get("ns"):
                raise XacroError("xacro:include ns= is not supported by rtbhip.xacro")
            if not os.path.isabs(fn):
                fn = os.path.join(os.path.dirname(os.path.abspath(self.files[-1])), fn)
            if not os.path.isfile(fn):
                if _boolean(attr.get("optional", "false"), "optional"):
                    return []
                raise XacroError("xacro:include: no such file %s (included from %s)" % (fn, self.files[-1]))
            self.files.append(fn)
            try:
                return self.children(self.load(fn), scope)
            finally:
                self.files.pop()
        if name == "insert_block":
            bname = attr.get("name")
            for key in ("**" + bname, "*" + bname, bname):
                if key in scope:
                    blk = scope[key]
                    if not isinstance(blk, _Block):
                        raise XacroError("insert_block: %r is not a block" % bname)
                    out = []
                    for e in blk.elements:
                        e = copy.deepcopy(e)
                        out.extend([e] if blk.evaluated else self.node(e, scope))
                    return out
            raise XacroError("insert_block: no block named %r" % bname)
        if name in ("element", "attribute"):
            raise XacroError("xacro:%s is not supported by rtbhip.xacro" % name)
        if name == "call":
            name = _fmt(self.text(attr.get("macro", ""), scope))
            attr = {k: v for k, v in attr.items() if k != "macro"}
        if name in self.macros:
            return self.call(self.macros[name], el, attr, scope)
        raise XacroError("unknown macro or xacro tag: xacro:%s (in %s)" % (name, self.files[-1] if self.files else "?"))

    def call(self, macro, el, attr, scope):
        inner = _Scope(self, scope)
        given = dict(attr)
        blocks = [c for c in el if isinstance(c.tag, str)]
        for nm, kind, has_default, default, forward in macro.params:
            if kind:
                if not blocks:
                    raise XacroError("macro %s: block parameter %s%s is missing" % (macro.name, kind, nm))
                b = blocks.pop(0)
                done = self.node(copy.deepcopy(b), scope)            # a block belongs to the caller: evaluated in the caller's scope
                if kind == "**":                                     # **name inserts what the element CONTAINS, *name the element itself
                    done = [c for d in done for c in list(d)]
                inner.define(kind + nm, _Block(done, evaluated=True))
                continue
            if nm in given:
                inner.define(nm, _literal(self.text(given.pop(nm), scope)))
            elif forward and nm in scope:
                inner.define(nm, scope[nm])
            elif has_default:
                inner.define(nm, _literal(self.text(default, inner)))
            else:
                raise XacroError("macro %s: parameter %r is missing" % (macro.name, nm))
        if given:
            raise XacroError("macro %s: unknown parameter(s) %s" % (macro.name, ", ".join(sorted(given))))
        if blocks:
            raise XacroError("macro %s: %d unused block(s)" % (macro.name, len(blocks)))
        return self.children(copy.deepcopy(macro.body), inner)


def _fmt(v):
    return v if isinstance(v, str) else str(v)        # (a boolean prints as Python prints it, "True": what the reference's tool writes too)


def _fmt_text(v):
    return None if v is None else _fmt(v)


def process(path, mappings=None, packages=None):
    """Expand the xacro file at `path` to URDF text.  `mappings`: values for `xacro:arg`s ($(arg name)); `packages`: where `$(find pkg)` looks --
    a {name: folder} dict or a list of folders that contain the description folders (default: the ancestors of the file itself)."""
    path = os.fspath(path)
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    ctx = _Context(mappings, packages)
    ctx.files.append(path)
    root = ctx.load(path)
    scope = _Scope(ctx)
    ns, name = _split(root.tag)
    if ns == "xacro":
        raise XacroError("the root element of %s is an xacro tag" % path)
    out = ET.Element(root.tag)
    for k, v in root.attrib.items():
        out.set(k, _fmt(ctx.text(v, scope)))
    out.text = root.text
    for c in ctx.children(root, scope):
        out.append(c)
    _indent(out)
    return '<?xml version="1.0" ?>\n' + ET.tostring(out, encoding="unicode") + "\n"


def _indent(el, level=0):
    pad = "\n" + "  " * level
    kids = list(el)
    if kids:
        if not (el.text or "").strip():
            el.text = pad + "  "
        for k in kids:
            _indent(k, level + 1)
            if not (k.tail or "").strip():
                k.tail = pad + "  "
        if not (kids[-1].tail or "").strip():
            kids[-1].tail = pad
    if level and not (el.tail or "").strip():
        el.tail = pad
