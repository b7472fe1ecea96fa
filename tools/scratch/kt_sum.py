import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
t = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")]
rows = db.execute(f"select s.string, count(*), avg(k.end - k.start) from {t[0]} k join {[r[0] for r in db.execute(chr(34).join(['select name from sqlite_master where type=','table',' and name like ','rocpd_info_kernel_symbol%','']).replace(chr(34), chr(39)))][0]} ks on k.kernel_id = ks.id join {[r[0] for r in db.execute('select name from sqlite_master where type=' + chr(39) + 'table' + chr(39) + ' and name like ' + chr(39) + 'rocpd_string%' + chr(39))][0]} s on ks.name_id = s.id group by s.string order by 3 desc").fetchall() if False else []
